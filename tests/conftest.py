"""pytest configuration: the `gpu` marker and shared loaders."""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
if str(ROOT / "tests") not in sys.path:
    sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _order_group(item):
    """Collection order of a `-x` run: the oracle-parity files first (test_gpu_00_configs: one test per BASELINE.json
    config; then parity, shapes, the random campaign, full sizes, host paths, slabs, replay), every other file after
    them, and the files that start child processes, launchers and rendezvous (test_gpu_9*: bench.py, torch.distributed.run)
    last -- an infrastructure test can then never stop the run before parity has been checked."""
    name = Path(str(item.fspath)).name
    if name.startswith("test_gpu_9"):
        return (2, name)
    if name.startswith("test_gpu_"):
        return (0, name)
    return (1, name)


def pytest_collection_modifyitems(session, config, items):
    items.sort(key=_order_group)          # stable: the order inside a file is kept


@pytest.fixture(scope="session")
def entry():
    import __graft_entry__ as g
    return g


@pytest.fixture(scope="session")
def pkg(entry):
    """The product package (HIP library behind the C-ABI).  Built on demand."""
    p = entry.PKG_DIR / "libamt_advance_mu_t.so"
    if not p.exists():
        entry.build()
    return entry.load_package()


@pytest.fixture(scope="session")
def oracle(entry):
    """TEST INFRASTRUCTURE: the CPU checker (oracle/)."""
    o = entry.load_oracle()
    o.lib()
    return o


def bits_equal(a, b) -> bool:
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8))


def slow_note(what: str, seconds: float, budget: float) -> None:
    """Wall-clock is never part of a parity verdict: a run that takes longer than its budget on a
    slow lease is reported -- printed and raised as a warning pytest lists in its summary -- and the test goes on."""
    import warnings
    print(f"  {what}: {seconds:.1f} s (budget {budget:.0f} s)")
    if seconds > budget:
        warnings.warn(f"{what} took {seconds:.1f} s, over its {budget:.0f} s budget (slow lease? not a parity failure)")
