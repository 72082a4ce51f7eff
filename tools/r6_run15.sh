#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r6; mkdir -p $O
export SM_BENCH_BACKEND=gloo MASTER_ADDR=127.0.0.1 OMP_NUM_THREADS=2 SM_FAULTHANDLER_S=240
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29591 bench.py --gpus 2 --steps 3 --warmup 1 > $O/b2_default.out 2> $O/b2_default.err; echo "rc $?" >> $O/b2_default.err
GPU_MAX_HW_QUEUES=4 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29593 bench.py --gpus 2 --steps 3 --warmup 1 > $O/b2_q4.out 2> $O/b2_q4.err; echo "rc $?" >> $O/b2_q4.err
grep -v "^\[W\|amdgpu.ids" $O/b2_default.err | tail -40; echo ======; grep -v "^\[W\|amdgpu.ids" $O/b2_q4.err | tail -15
