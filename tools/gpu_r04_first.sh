#!/bin/bash
# GPU box, round 4, first measurements on the XCD-banded attention: default bench line, SQ counters on the attention kernels, the serial
# kernel profile at 4 utterances per GPU (the per-rank batch at N = 8) and at batch 32.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 400 python3 bench.py > gpurun_out/r04_bench_b32_v1.json 2> gpurun_out/r04_bench_b32_v1.err; echo "bench exit $?"; tail -n 1 gpurun_out/r04_bench_b32_v1.json | cut -c1-1500
tools/gpu_pmc_attn.sh r04_attn_xcd > gpurun_out/r04_attn_xcd_pmc.log 2>&1; echo "pmc exit $?"; tail -n 30 gpurun_out/r04_attn_xcd_pmc.log
STEPS=3 tools/gpu_prof.sh r04_b4_serial --profile-serial --global-batch 4 | head -45
STEPS=3 tools/gpu_prof.sh r04_b32_serial --profile-serial | head -45
