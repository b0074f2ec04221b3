cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_model_gpu.py -x -q -s -k "sharded_optimizer or graphed_ddp or fused_adamw or rccl_reducer or grad_accum or best_pt" > gpurun_out/r04_shard_test.txt 2>&1; echo "pytest exit $?"; grep -n "sharded optimizer:\|passed\|failed\|Error" gpurun_out/r04_shard_test.txt | cut -c1-700
tools/gpu_r04_shard_ab.sh
