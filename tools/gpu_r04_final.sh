#!/bin/bash
# GPU box: the measurements that back round 4's numbers -> gpurun_out/ (copied into profiles/ afterwards).
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_smoke.log 2>&1; echo "smoke exit $?"; grep "smoke\[" gpurun_out/r04_smoke.log
timeout -k 10 500 python3 bench.py > gpurun_out/r04_bench_b32_default.json 2> gpurun_out/r04_bench_b32_default.err; echo "bench exit $?"; tail -n 1 gpurun_out/r04_bench_b32_default.json | cut -c1-300
HEAD_SHA=$1 tools/gpu_round_profiles.sh 2>&1 | tail -n 90
timeout -k 10 200 python3 bench.py --global-batch 4 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/r04_bench_b4_single_graph.json 2>/dev/null; tail -n 1 gpurun_out/r04_bench_b4_single_graph.json | cut -c1-260
TAV_B=32 timeout -k 10 200 python3 tools/gpu_ab.py layer attn tng > gpurun_out/r04_microbench_b32.txt 2>&1; grep "total\|video" gpurun_out/r04_microbench_b32.txt | head -20
TAV_BENCH_REHEARSE=1 timeout -k 10 400 python3 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-roofline > gpurun_out/r04_bench_rehearsal_2ranks_one_gpu.json 2> gpurun_out/r04_rehearsal.err; echo "rehearsal exit $?"; tail -n 1 gpurun_out/r04_bench_rehearsal_2ranks_one_gpu.json | cut -c1-260
