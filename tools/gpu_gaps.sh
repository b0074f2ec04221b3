#!/bin/bash
# GPU box: kernel trace of a few replayed steps of the default bench -> idle time of the device and of the busiest queue (where do the ms between
# "sum of kernel times" and "wall" go?).  usage: tools/gpu_gaps.sh [extra bench.py arguments, e.g. --global-batch 4]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gaps -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline "$@" > gpurun_out/gaps.json 2> gpurun_out/gaps.err
echo "rocprof exit $?"
f=$(find gpurun_out/gaps -name "*kernel_trace.csv" | head -1)
python3 tools/trace_gaps.py "$f" | tee gpurun_out/gaps_summary.txt
find gpurun_out/gaps -type f -delete 2>/dev/null
