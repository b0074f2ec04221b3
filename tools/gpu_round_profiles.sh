#!/bin/bash
# GPU box: the measurements that back the round's numbers -> gpurun_out/ (copy what is quoted into profiles/).
#   1. the data-parallel bench path with ONE rank on the real RCCL backend (ddp.GraphedStep)
#   2. rocprofv3 kernel trace + stats of the default bench command and of its --profile-serial form (each kernel alone on the device)
#   3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (counters only with --kernel-trace) -> traffic_per_launch.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
TAV_DDP_SINGLE_RANK=1 timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/bench_ddp1.json 2> gpurun_out/bench_ddp1.err
echo "ddp single-rank bench exit $?"; tail -n 1 gpurun_out/bench_ddp1.json | cut -c1-700; grep "hipGraphs\|timed region\|rror" gpurun_out/bench_ddp1.err | head -5
STEPS=3 tools/gpu_prof.sh r04_b32_serial --profile-serial | head -40
STEPS=5 tools/gpu_prof.sh r04_b32_default | head -12
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary > gpurun_out/pmc_$c.log 2>&1 || { echo "pmc pass $c failed"; tail -n 5 gpurun_out/pmc_$c.log; exit 1; }
done
python3 tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/traffic_per_launch.json --batch 32 --preset B --head "${HEAD_SHA:-}" --steps-in-trace 5 --skip-steps 1
# raw per-dispatch counter CSVs (compressed) -> copy to profiles/<round>_pmc/ beside traffic_per_launch.json
for c in FETCH_SIZE WRITE_SIZE; do f=$(find gpurun_out/pmc_$c -name "*counter_collection.csv" | head -1); gzip -c "$f" > gpurun_out/pmc_${c}_counter_collection.csv.gz; done
find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -type f -delete 2>/dev/null
