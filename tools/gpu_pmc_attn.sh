#!/bin/bash
# GPU box: rocprofv3 --pmc passes over the attention kernels at the benchmarked video shape (tools/pmc_attn.py).  Counters only with
# --kernel-trace (never with the sys/hip/hsa trace domains).  Raw CSVs are kept under gpurun_out/<tag>/ ; copy them to profiles/<round>_pmc/.
#   usage: tools/gpu_pmc_attn.sh <tag> [target.py] [kernel-name filter]     (TAV_LIB selects an A/B build; defaults: tools/pmc_attn.py, "attn_")
tag=${1:-pmc_attn}
target=${2:-tools/pmc_attn.py}
export TAV_PMC_FILTER=${3:-attn_}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; mkdir -p $out
n=0
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VALU SQ_INST_CYCLES_SALU" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM SQ_INSTS_VALU_CVT SQ_LDS_DATA_FIFO_FULL"; do
  n=$((n+1))
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pass$n -- python3 $target > $out/pass$n.log 2>&1 || { echo "pass $n failed"; tail -n 5 $out/pass$n.log; exit 1; }
  f=$(find $out/pass$n -name "*counter_collection.csv" | head -1)
  cp "$f" $out/pass${n}_counter_collection.csv
  find $out/pass$n -type f -delete 2>/dev/null
done
python3 tools/pmc_attn_summary.py $out
