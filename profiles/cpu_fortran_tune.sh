#!/bin/bash
# A/B of the Fortran CPU path's build knobs ON THE TIMING HOST (the GPU box's EPYC cores; no GPU is touched):
# columns per i block (AMT_IB) and the loop order (AMT_J_OUTER=1: r03's j-outer order), each built
# -march=native into its own directory and timed by oracle/cpu_bench.py on all granted cores and on one.
#   bash profiles/cpu_fortran_tune.sh [out-dir]
O=${1:-gpurun_out/cpu_tune}
mkdir -p $O
python3 oracle/cpu_bench.py --prebuild > $O/prebuild.json 2>&1
CORES=$(python3 -c "import bench; print(bench.host_cores()[0])")
echo "cores $CORES; $(grep -m1 'model name' /proc/cpuinfo)" | tee $O/summary.txt
for impl in c; do
  for sz in "4096 60 256 $CORES" "512 60 512 1"; do
    set -- $sz
    echo "port_c $1x$2x$3 x$4: $(python3 oracle/cpu_bench.py --impl c --size $1 $2 $3 --threads $4 --seconds 2 | tail -1)" | tee -a $O/summary.txt
  done
done
for order in 0 1; do
  for ib in 64 128 256 512 1024 4096; do
    D=/tmp/fcpu_${order}_${ib}
    mkdir -p $D
    make -s -C oracle FCPU_OUT=$D FNATIVE=-march=native FCPU_DEFS="-DAMT_IB=$ib -DAMT_J_OUTER=$order" fortran_cpu > $D/build.log 2>&1 || { echo "build failed ib=$ib order=$order"; tail -3 $D/build.log; continue; }
    for sz in "4096 60 256 $CORES" "512 60 512 1"; do
      set -- $sz
      echo "fortran j_outer=$order ib=$ib $1x$2x$3 x$4: $(AMT_FORTRAN_CPU_DIR=$D python3 oracle/cpu_bench.py --impl fortran --size $1 $2 $3 --threads $4 --seconds 2 | tail -1)" | tee -a $O/summary.txt
    done
  done
done
# fp32 at the configs[4] row shape with the two best candidates
for ib in 256 512 1024; do
  D=/tmp/fcpu_0_${ib}
  echo "fortran f32 ib=$ib 8192x80x128 x$CORES: $(AMT_FORTRAN_CPU_DIR=$D python3 oracle/cpu_bench.py --impl fortran --dtype f32 --size 8192 80 128 --threads $CORES --seconds 2 | tail -1)" | tee -a $O/summary.txt
done
echo "port_c f32 8192x80x128 x$CORES: $(python3 oracle/cpu_bench.py --impl c --dtype f32 --size 8192 80 128 --threads $CORES --seconds 2 | tail -1)" | tee -a $O/summary.txt
