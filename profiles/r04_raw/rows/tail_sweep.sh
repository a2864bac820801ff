python - <<'PY'
# parity of the ticketed tail first: small domain, forced one-round plan
import sys; sys.path.insert(0, '.')
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package(); oracle = g.load_oracle(); L = pkg.load_library(); S = pkg.synth
for dtype in (np.float64, np.float32):
    for (ni, nk, nj, tr, ts) in [(1024, 20, 300, 8, 2), (2048, 12, 200, 16, 4), (700, 30, 130, 5, 3), (4096, 8, 100, 7, 1)]:
        b = S.domain_bounds(ni, nk, nj)
        host = S.make_patch(b, pkg.GridConfig(specified=True), dtype=dtype, seed=11)
        want = host.copy(); oracle.advance_mu_t(*want.args())
        L.amt_march_set_tail(tr, ts)
        for rep in range(3):
            dev = host.to_device("cuda:0")
            pkg.advance_mu_t(*dev.args(), variant=pkg.VARIANT_MARCH)
            torch.cuda.synchronize()
            got = dev.to_host()
            bad = [n for n in S.OUTPUTS if not np.array_equal(got.arrays[n].view(np.uint8), want.arrays[n].view(np.uint8))]
            print(np.dtype(dtype).name, ni, nk, nj, tr, ts, L.amt_march_last_kernel().decode()[-40:], "DIFF " + str(bad) if bad else "ok")
L.amt_march_set_tail(0, 4)
PY
for cfg in "--dtype f64 --ni 4096 --nk 60 --nj 4096" "--dtype f32 --ni 4096 --nk 60 --nj 4096" "--dtype f64 --ni 4096 --nk 60 --nj 512" "--dtype f64 --ni 2048 --nk 60 --nj 2048"; do
 for rep in 1 2; do python profiles/rows_sweep.py $cfg --rows 0 --tail 0/4,16/4,32/4,32/8,64/8,64/16,24/2 --rounds 4 2>&1 | grep -v amdgpu.ids; done
done
