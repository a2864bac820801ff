#!/bin/bash
# profiles/size_sweep.sh -- bench.py over a range of shapes (GPU box); one JSON line per shape
# into gpurun_out/size_sweep.jsonl (summarised into profiles/r01_size_sweep.md by hand).
OUT=gpurun_out/size_sweep.jsonl; : > $OUT
run() { timeout 600 python bench.py --no-cpu-baseline --steps $1 --warmup 3 --dtype $2 --ni $3 --nk $4 --nj $5 2>/dev/null | tail -1 >> $OUT; }
run 500 f64 64 40 64
run 500 f64 128 60 128
run 300 f64 256 60 256
run 200 f64 512 60 512
run 100 f64 1024 60 1024
run 50 f64 2048 60 2048
run 30 f64 4096 60 4096
run 100 f64 4096 60 512
run 30 f64 4096 40 4096
run 30 f64 4096 20 4096
run 30 f64 4096 76 2048
run 200 f32 512 60 512
run 30 f32 4096 60 4096
run 30 f32 8192 80 2048
run 10 f32 8192 80 8192
python - <<'PY'
import json
print("| shape | dtype | ms/sweep | Gcells/s | algorithmic GB/s | of 8 TB/s |")
print("|---|---|---|---|---|---|")
for l in open("gpurun_out/size_sweep.jsonl"):
    d = json.loads(l); c = d["config"]
    print(f"| {c['ni']}×{c['nk']}×{c['nj']} | {d['dtype']} | {d['ms_per_step']:.4f} | {d['value']/1e3:.1f} | {d['roofline']['achieved']:.0f} | {100*d['roofline']['frac']:.1f} % | verified {d.get('verified_vs_oracle')} |")
PY
