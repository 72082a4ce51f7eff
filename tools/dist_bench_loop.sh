#!/bin/bash
# Repeat the 2-ranks-on-one-GPU (gloo) bench run N times; a stalled run dumps every rank's stacks (SM_FAULTHANDLER_S) and is
# killed as a process group.  usage (through gpurun): bash tools/dist_bench_loop.sh [N]
n=${1:-10}; out=$PWD/gpurun_out/dist_loop; mkdir -p $out
for i in $(seq 1 $n); do
  SM_BENCH_BACKEND=gloo MASTER_ADDR=127.0.0.1 OMP_NUM_THREADS=2 SM_FAULTHANDLER_S=60 timeout -s KILL 100 setsid \
    python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port $((29600 + i)) \
    bench.py --gpus 2 --steps 3 --warmup 1 > $out/run$i.log 2>&1
  rc=$?
  echo "run $i rc=$rc $(grep -c '^{' $out/run$i.log) json line(s)"
  if [ $rc -ne 0 ]; then tail -60 $out/run$i.log; fi
done
