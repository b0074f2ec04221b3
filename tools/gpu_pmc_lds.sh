#!/bin/bash
# GPU box: LDS bank-conflict counters of the GEMM kernels on the microbench (tools/gpu_ab.py <sections>); prints conflict / active cycles per kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_lds -- python3 tools/gpu_ab.py "$@" > gpurun_out/pmc_lds.log 2>&1
echo "rocprof exit $?"
f=$(find gpurun_out/pmc_lds -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    agg[r["Kernel_Name"][:90]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(agg.items()):
    if "tav::" in k:
        a, c = v.get("SQ_LDS_IDX_ACTIVE", 0.0), v.get("SQ_LDS_BANK_CONFLICT", 0.0)
        print("%-90s conflict/active = %.3f  (active %.3e)" % (k, c / a if a else 0.0, a))
PY
find gpurun_out/pmc_lds -type f -delete 2>/dev/null
