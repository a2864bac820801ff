#!/bin/bash
# profiles/quick.sh "ENV=.. ENV=.." ... -- one bench line per environment setting (GPU box helper)
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['verified_vs_oracle'])"
done
