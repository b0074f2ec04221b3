cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for v in "" "--shard-optimizer" "" "--shard-optimizer"; do
  TAV_DDP_SINGLE_RANK=1 timeout -k 10 200 python3 bench.py --global-batch 4 --steps 30 --warmup 3 --no-cpu-baseline --no-secondary --no-roofline $v 2> gpurun_out/r04_shard_b4.err | tail -n 1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('b4 one rank on RCCL [$v]:', d['ms_per_step'], 'ms/step', d['value'], d['unit'])" || exit 1
done
