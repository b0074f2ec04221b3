#!/bin/bash
# GPU box: the data-parallel step with ONE rank on real RCCL -- graph chain (rounds 2-3) against the single graph with captured raw RCCL
# all-reduces (round 4) -- at 4 utterances per GPU (the per-rank batch at N = 8) and at batch 32, next to the plain single-GPU graph.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 400 python3 -m pytest tests/test_model_gpu.py tests/test_00_ddp_two_ranks_gpu.py -x -q -k "graphed_ddp or rccl or c_abi or two_ranks" -s > gpurun_out/r04_ddp_tests.log 2>&1; rc=$?
echo "ddp tests exit $rc"; grep -a "passed\|failed\|graphed ddp:\|RCCL runtime" gpurun_out/r04_ddp_tests.log | cut -c1-400
[ $rc -eq 0 ] || { tail -n 40 gpurun_out/r04_ddp_tests.log; exit 1; }
for gb in 4 32; do
  tools/bench_line.sh "b$gb single GPU, one graph (no reducer)        " --global-batch $gb
  TAV_DDP_SINGLE_RANK=1 tools/bench_line.sh "b$gb 1-rank RCCL, graph chain                 " --global-batch $gb --ddp-mode chain
  TAV_DDP_SINGLE_RANK=1 tools/bench_line.sh "b$gb 1-rank RCCL, single graph, captured RCCL " --global-batch $gb --ddp-mode single
  TAV_DDP_SINGLE_RANK=1 TAV_DDP_SKIP_REDUCE=1 tools/bench_line.sh "b$gb single graph, collectives skipped        " --global-batch $gb --ddp-mode single
done 2>&1 | tee gpurun_out/r04_ddp_ab.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/nb -- python3 bench.py --steps 1 --warmup 1 --profile-serial --no-cpu-baseline --no-secondary --no-roofline --global-batch 4 > gpurun_out/nb.json 2> gpurun_out/nb.err
f=$(find gpurun_out/nb -name "*kernel_trace.csv" | head -1)
python3 tools/trace_neighbours.py "$f" | tee gpurun_out/r04_fill_neighbours.txt | head -45
find gpurun_out/nb -type f -delete 2>/dev/null
