#!/bin/bash
# GPU box: kernel checks, then same-box A/B of the round-4 residue work (deferred LayerNorm parameter reduce, phase-GEMM conv input
# gradients) on the default bench at batch 32 and at 4 utterances per GPU, then the ATen launch audit.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 500 python3 tools/gpu_kernel_check.py > gpurun_out/kernel_check.log 2>&1; rc=$?; echo "kernel checks exit $rc"; grep -c "^ok" gpurun_out/kernel_check.log; grep "FAIL\|EXC\|failures\|Error" gpurun_out/kernel_check.log | head -20
[ $rc -eq 0 ] || { tail -n 30 gpurun_out/kernel_check.log; exit 1; }
for gb in 32 4; do
  TAV_LN_DEFER=0 TAV_CONV_PHASE=0 tools/bench_line.sh "b$gb baseline (immediate LN reduce, col2im)" --global-batch $gb
  TAV_LN_DEFER=1 TAV_CONV_PHASE=0 tools/bench_line.sh "b$gb deferred LN reduce             " --global-batch $gb
  TAV_LN_DEFER=0 TAV_CONV_PHASE=1 tools/bench_line.sh "b$gb phase-GEMM conv dgrad          " --global-batch $gb
  TAV_LN_DEFER=1 TAV_CONV_PHASE=1 tools/bench_line.sh "b$gb both                           " --global-batch $gb
done 2>&1 | tee gpurun_out/r04_residue_ab.txt
TAV_B=4 timeout -k 10 300 python3 tools/gpu_aten_audit2.py > gpurun_out/r04_aten_audit_b4.txt 2>&1; echo "audit exit $?"; tail -n 45 gpurun_out/r04_aten_audit_b4.txt
