#!/bin/bash
# GPU box: the sharded optimizer (optim.ShardedAdamW) behind bench.py -- ONE rank on real RCCL at 4 utterances (what the extra graph boundaries and the
# wire copies cost when a rank owns everything), and the two-rank rehearsal on one GPU over gloo (that the bench path runs end to end).
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for v in "" "--shard-optimizer" "" "--shard-optimizer"; do
  TAV_DDP_SINGLE_RANK=1 timeout -k 10 200 python3 bench.py --global-batch 4 --steps 30 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline $v 2> gpurun_out/r04_shard_b4.err | tail -n 1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('b4 one rank on RCCL [$v]:', d['ms_per_step'], 'ms/step', d['value'], d['unit'], '|', d['config'].get('launch','')[:160])" || exit 1
done
TAV_BENCH_REHEARSE=1 timeout -k 10 400 python3 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-roofline --shard-optimizer > gpurun_out/r04_bench_rehearsal_2ranks_sharded.json 2> gpurun_out/r04_rehearsal_sharded.err; echo "rehearsal exit $?"; tail -n 1 gpurun_out/r04_bench_rehearsal_2ranks_sharded.json | cut -c1-900
grep -n "SHARDED" gpurun_out/r04_rehearsal_sharded.err | cut -c1-400
