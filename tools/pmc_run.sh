#!/bin/bash
# usage (GPU box): tools/pmc_run.sh <outdir> "<counters pass 1>" "<counters pass 2>" ...   -- one rocprofv3 --pmc pass per argument
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
n=0
for c in "$@"; do
  n=$((n+1))
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out/pass$n" -- python3 tools/pmc_target.py > "$out.pass$n.log" 2>&1 || { echo "pass $n failed"; tail -5 "$out.pass$n.log"; exit 1; }
done
