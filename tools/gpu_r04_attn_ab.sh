#!/bin/bash
# GPU box, one call: kernel checks on the default library, then the XCD-local attention grid (build/ab/xcd1.so) against the plain grid
# (build/ab/xcd0.so) on the SAME box: timings (tools/gpu_ab.py attn, batch 32) and HBM traffic (tools/gpu_attn_traffic.sh).
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 500 python3 tools/gpu_kernel_check.py > gpurun_out/kernel_check.log 2>&1; echo "kernel checks exit $?"; grep -c "^ok" gpurun_out/kernel_check.log; grep "FAIL\|EXC\|failures" gpurun_out/kernel_check.log
for v in xcd0 xcd1; do
  TAV_B=32 TAV_LIB=build/ab/$v.so timeout -k 10 200 python3 tools/gpu_ab.py attn > gpurun_out/ab_attn_$v.txt 2>&1 || exit 1
  grep "pre1" gpurun_out/ab_attn_$v.txt
done
for v in xcd0 xcd1; do
  TAV_LIB=build/ab/$v.so tools/gpu_attn_traffic.sh attn_traffic_$v || exit 1
done
