"""TEST INFRASTRUCTURE: runs bench.py's --cpu-dry-run with the ORACLE injected as the compute callable (the product has no
CPU path; bench.py refuses the dry run without an injection).  Started by tests/test_bench_dry_run.py as
`python tests/workers/bench_cpu_shim.py --gpus 8 --cpu-dry-run ...`; bench.py's self-launcher starts its ranks as this same
script, so every rank gets the injection."""
import importlib.util
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))

spec = importlib.util.spec_from_file_location("amt_bench", ROOT / "bench.py")
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def _oracle_compute(*args):
    import torch
    import __graft_entry__ as g
    g.load_oracle().advance_mu_t(*[x.numpy() if isinstance(x, torch.Tensor) else x for x in args])


bench.DRY_RUN_COMPUTE = _oracle_compute
if "--cpu-dry-run" not in sys.argv:
    raise SystemExit("bench_cpu_shim.py only serves --cpu-dry-run")
bench.main()
