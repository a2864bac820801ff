# FETCH_SIZE of the fp64 80-level launch with 64-row and 128-row blocks, same box
export TMPDIR=/tmp
O=gpurun_out/fetch80; mkdir -p $O
A="--nk 80 --nj 2048 --steps 4 --warmup 1 --no-cpu-baseline --no-verify --no-traffic --no-box-probe --probe-placements 1"
for r in 64 128 32; do
  export AMT_MARCH_JROWS=$r
  rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $O/f$r -o f -- python3 bench.py $A > $O/bench_$r.log 2>&1
done
unset AMT_MARCH_JROWS
python3 - <<'PY'
import csv, glob
for r in (64, 128, 32):
    for f in glob.glob(f"gpurun_out/fetch80/f{r}/**/*counter_collection.csv", recursive=True):
        v = [float(x["Counter_Value"]) for x in csv.DictReader(open(f)) if "amt_march_kernel" in x["Kernel_Name"] and x["Counter_Name"] == "FETCH_SIZE"]
        print(r, len(v), sum(v) / len(v) * 2 * 1024 / 1e9, "GB read per launch")
PY
