#!/bin/bash
# profiles/r06_ipc_stress.sh -- GPU box: the soak of the IPC mailbox protocol, now able to SEE a stale or torn delivery (VERDICT r05
# item 1): 8 processes on the one GPU, 2000 single-sweep calls each with a random host sleep of up to 200 us in front of every call;
# every sweep sends DIFFERENT rows (the exchanged fields refilled with seed + sweep: tests/workers/slab_ipc_rank.py) into NaN-poisoned
# halos; host-waited and device-waited schedules, kernel and copy-engine pulls; the result after 2000 sweeps must be the unsplit
# oracle run's, bit for bit (the checker refills the whole domain the same way before each of its sweeps).
set -u
cd ${GRAFT_REPO_ROOT:-.}
for MODE in "1 kernel" "0 kernel" "1 engine" "0 engine"; do
  set -- $MODE; HW=$1; PULL=$2
  D=$(mktemp -d /tmp/amt_stress_XXXX)
  for r in 0 1 2 3 4 5 6 7; do
    AMT_RENDEZVOUS_NONCE=stress-$HW-$PULL AMT_SLAB_TRANSPORT=ipc AMT_IPC_HOST_WAIT=$HW AMT_IPC_PULL=$PULL AMT_IPC_DEVICE_TIMEOUT_S=20 AMT_IPC_TIMEOUT_S=120 HSA_ENABLE_IPC_MODE_LEGACY=0 \
      python3 tests/workers/slab_ipc_rank.py --rank $r --world 8 --dir $D --dims 128 16 64 --sweeps 2000 --jitter-us 200 > $D/log_$r.txt 2>&1 &
  done
  wait
  echo "host_wait=$HW pull=$PULL: $(grep -l 'ranks seen 8' $D/log_*.txt | wc -l) of 8 ranks finished ($(grep -h -o 'inputs new every sweep' $D/log_0.txt | head -1))"; grep -h "Error\|error" $D/log_*.txt | head -3
  python3 - $D <<'PY'
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import __graft_entry__ as g
from multirank import slab_mismatches
from pathlib import Path
pkg, oracle = g.load_package(), g.load_oracle()
bad = slab_mismatches(pkg, oracle, Path(sys.argv[1]), 8, (128, 16, 64), "f64", 2000)
print("   (rank, array) pairs differing from the unsplit oracle run after 2000 sweeps with new inputs every sweep:", bad)
PY
  rm -rf $D
done
