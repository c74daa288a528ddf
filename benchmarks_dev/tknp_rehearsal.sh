#!/bin/bash
# Token-parallel micro-benchmark with N ranks on ONE GPU (rehearsal: all ranks share cuda:0, so the numbers
# show launch / exchange overhead, not multi-GPU speed-up).  usage: tknp_rehearsal.sh N [extra bench args]
N=$1; shift
mkdir -p gpurun_out
for r in $(seq 0 $((N-1))); do
  RANK=$r LOCAL_RANK=$r WORLD_SIZE=$N MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 HSA_ENABLE_IPC_MODE_LEGACY=0 \
  VDA_REHEARSE_ON_ONE_GPU=1 VLLM_LOGGING_LEVEL=WARNING \
  timeout -k 10 400 python -u -m vllm_distributed_amd.benchmarks.tknp --token-parallel-size $N \
    --gpu-memory-utilization 0.2 "$@" > gpurun_out/tknp_rank$r.log 2>&1 &
done
wait
grep -v "amdgpu.ids\|Gloo\|socket" gpurun_out/tknp_rank0.log | tail -12
