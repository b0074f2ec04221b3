"""Full-depth bf16 parity of a preset on several seeds (weights and inputs), using the GPU test suite's own helpers.
usage: python tools/gpu_seed_probe.py A 0 1 2"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch  # noqa: E402

torch.set_num_threads(16)       # the CPU oracle: the cgroup's threads, not the host's core count
import test_model_gpu as T  # noqa: E402

preset = sys.argv[1]
for seed in [int(a) for a in sys.argv[2:]]:
    try:
        errs = T._compare_with_oracle(T._oracle_full(preset, seed=seed), "bf16", 1.0, f"probe {preset} seed {seed}", grad_tol=1.0, check_optimizer=False)
        print(f"preset {preset} seed {seed}: logits {errs[0]:.3e} loss {errs[1]:.3e} grad-norm {errs[2]:.3e} worst tensor {errs[3]:.3e}", flush=True)
    except AssertionError as e:
        print(f"preset {preset} seed {seed}: assertion {str(e)[:200]}", flush=True)
