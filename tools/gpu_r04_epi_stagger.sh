#!/bin/bash
# GPU box: register epilogue (plain / residual flavours) x CUs out of step inside every XCD, on the video layer's NT GEMMs (build/direct_src: scratch sources)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; out=gpurun_out/r04_epi_stagger.txt; : > $out
for v in staged direct staged_st50 direct_st50 direct_st100 staged direct_st50; do
  TAV_B=32 TAV_LIB=build/ab/$v.so timeout -k 10 120 python tools/gpu_ab.py layer >> $out 2>&1 || exit 1
done
echo "---- one tile per CU (batch 1, 256 x 256 tiles forced): the epilogue's cost on an otherwise idle chip" >> $out
for v in staged direct; do
  TAV_B=1 TAV_TM=16 TAV_LIB=build/ab/$v.so timeout -k 10 120 python tools/gpu_ab.py layer >> $out 2>&1 || exit 1
done
grep -v amdgpu.ids $out
