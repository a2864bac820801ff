#!/bin/bash
# r03 pass 5 (GPU box): the new bench line, failure path of N > 1, whole GPU suite
set -u
O=gpurun_out/diag5; mkdir -p $O
python3 bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "rc $?" >> $O/bench_default.err
python3 bench.py --gpus 2 --backend gloo --ni 512 --nj 512 --steps 3 --warmup 2 > $O/bench_gloo2.json 2> $O/bench_gloo2.err; echo "rc $?" >> $O/bench_gloo2.err
( time python3 bench.py --gpus 2 --share-gpu --comm-timeout 40 --launch-timeout 300 --ni 512 --nj 512 --steps 3 --warmup 2 ) > $O/bench_share.json 2> $O/bench_share.err; echo "rc $?" >> $O/bench_share.err
timeout 2400 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -c 3000 $O/bench_default.json; tail -5 $O/bench_default.err; tail -20 $O/bench_share.err; tail -5 $O/pytest.log
