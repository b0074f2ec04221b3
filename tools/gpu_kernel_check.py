"""Run every kernel check and print a table (does not stop at the first failure).  GPU box only."""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch  # noqa: E402

import kernel_checks  # noqa: E402

bad = 0
for i, c in enumerate(kernel_checks.all_checks()):
    try:
        for name, err, tol, ok in c():
            print(f"{'ok  ' if ok else 'FAIL'} {name:70s} err={err:.3e} tol={tol:.1e}", flush=True)
            bad += 0 if ok else 1
        torch.cuda.synchronize()
    except Exception:
        bad += 1
        print(f"EXC  case {i}", flush=True)
        traceback.print_exc()
print("failures:", bad)
sys.exit(1 if bad else 0)
