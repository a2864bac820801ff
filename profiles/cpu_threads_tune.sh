#!/bin/bash
# Thread placement / wait policy of the two CPU paths on the timing host (no GPU is touched): the cgroup grants 16 of 256
# logical CPUs, and where the 16 threads land (CCDs, sockets) and how idle threads wait moves the figure more than the code does.
O=${1:-gpurun_out/cpu_threads}
mkdir -p $O
python3 oracle/cpu_bench.py --prebuild > $O/prebuild.json 2>&1
CORES=$(python3 -c "import bench; print(bench.host_cores()[0])")
echo "cores $CORES; $(grep -m1 'model name' /proc/cpuinfo); $(nproc) nproc; affinity $(python3 -c 'import os; print(len(os.sched_getaffinity(0)))')" | tee $O/summary.txt
lscpu | grep -E "NUMA|Socket|L3" | tee -a $O/summary.txt
run() {  # label, env..., --, args
  label=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  echo "$label $*: $(env "${envs[@]}" python3 oracle/cpu_bench.py "$@" --seconds 2 | tail -1)" | tee -a $O/summary.txt
}
for rep in 1 2; do
for impl in c fortran; do
  for sz in "4096 60 256" "512 60 512"; do
    run "default" X=1 -- --impl $impl --size $sz --threads $CORES
    run "spread/cores" OMP_PROC_BIND=spread OMP_PLACES=cores -- --impl $impl --size $sz --threads $CORES
    run "close/cores" OMP_PROC_BIND=close OMP_PLACES=cores -- --impl $impl --size $sz --threads $CORES
    run "passive" OMP_WAIT_POLICY=passive -- --impl $impl --size $sz --threads $CORES
    run "active" OMP_WAIT_POLICY=active -- --impl $impl --size $sz --threads $CORES
  done
done
done
