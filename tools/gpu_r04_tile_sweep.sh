#!/bin/bash
# GPU box: does the NT tile model (nt_pick_tile, hint 0) still pick the fastest tile for every stack's row count after rounds 3-4 changed the kernels?
# One layer's eight NT GEMMs with their real epilogues, batch 32, per stack: automatic choice vs every forced tile (2/3/4 = 64/96/128 x 128, 8 = 256 x 128, 16 = 256 x 256).
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; out=gpurun_out/r04_tile_sweep.txt; : > $out
for S in 481 249 128 1464; do
  for tm in 0 2 3 4 8 16; do
    echo "== tokens per utterance $S, tile hint $tm" >> $out
    TAV_B=32 TAV_S=$S TAV_TM=$tm timeout -k 10 120 python tools/gpu_ab.py layer 2>&1 | grep "layer-NT" | sed "s/^\[libtavhip.so\] //" >> $out || exit 1
  done
done
grep "==\|total" $out | paste - - | sed "s/layer-NT total//"
