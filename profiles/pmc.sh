#!/bin/bash
# profiles/pmc.sh TAG "COUNTER ..." [bench args] -- one extra rocprofv3 --pmc pass (GPU box helper);
# prints per-kernel means of each counter.
TAG=$1; CNT=$2; shift 2
ARGS=${@:---steps 3 --warmup 1 --no-cpu-baseline --no-verify}
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --pmc $CNT -d $OUT -o pmc -- python3 bench.py $ARGS > $OUT/bench.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if "march" in k or "column" in k:
        print(k)
        for c, v in cs.items():
            print(f"   {c:28s} {sum(v)/len(v):.4g}  (n={len(v)})")
PY
