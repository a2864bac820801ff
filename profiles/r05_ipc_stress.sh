set -u
cd $GRAFT_REPO_ROOT
for HW in 1 0; do
  D=$(mktemp -d /tmp/amt_stress_XXXX)
  for r in 0 1 2 3 4 5 6 7; do
    AMT_RENDEZVOUS_NONCE=stress-$HW AMT_SLAB_TRANSPORT=ipc AMT_IPC_HOST_WAIT=$HW AMT_IPC_DEVICE_TIMEOUT_S=20 AMT_IPC_TIMEOUT_S=120 HSA_ENABLE_IPC_MODE_LEGACY=0 \
      python tests/workers/slab_ipc_rank.py --rank $r --world 8 --dir $D --dims 128 16 64 --sweeps 2000 --jitter-us 200 > $D/log_$r.txt 2>&1 &
  done
  wait
  echo "host_wait=$HW: $(grep -l 'ranks seen 8' $D/log_*.txt | wc -l) of 8 ranks finished"; grep -h "Error\|error" $D/log_*.txt | head -3
  python - $D <<'PY'
import sys, numpy as np
sys.path.insert(0, ".")
import __graft_entry__ as g
pkg, oracle = g.load_package(), g.load_oracle()
S = pkg.synth
dims, world, sweeps = (128, 16, 64), 8, 2000
gb = S.domain_bounds(*dims, aligned=True)
want = S.make_patch(gb, pkg.GridConfig(), dtype=np.float64, seed=11, global_dims=dims)
for _ in range(sweeps):
    oracle.advance_mu_t(*want.args())
bad = 0
for r in range(world):
    sb = S.slab_bounds(gb, r, world)
    for n in S.OUTPUTS:
        got = np.load(f"{sys.argv[1]}/out_{r}_{n}.npy")
        w = want.arrays[n][sb.jts - gb.jms: sb.jte + 1 - gb.jms]
        if not np.array_equal(got.view(np.uint8), w.view(np.uint8)): bad += 1
print("arrays differing from the unsplit oracle after 2000 sweeps:", bad, "finite:", bool(np.isfinite(want.arrays["t"]).all()))
PY
done
