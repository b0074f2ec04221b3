#!/bin/bash
# GPU box: HBM traffic of the attention kernels at the video shape (tools/pmc_attn.py: B 32, S 1464, 12 heads), FETCH_SIZE / WRITE_SIZE in
# separate rocprofv3 --pmc passes (counters only with --kernel-trace).  usage: tools/gpu_attn_traffic.sh <tag>   (TAV_LIB selects an A/B build)
tag=${1:-attn_traffic}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$c -- python3 tools/pmc_attn.py > $out/$c.log 2>&1 || { echo "pass $c failed"; tail -n 5 $out/$c.log; exit 1; }
  f=$(find $out/$c -name "*counter_collection.csv" | head -1)
  cp "$f" $out/${c}_counter_collection.csv
  find $out/$c -type f -delete 2>/dev/null
done
python3 - $out <<'PY'
import csv, sys, collections, statistics
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for r in csv.DictReader(open(f"{out}/{c}_counter_collection.csv")):
        k = r["Kernel_Name"]
        if "attn_" not in k or r["Counter_Name"] != c:
            continue
        agg[k.split("(")[0].replace("void tav::", "")][c].append(float(r["Counter_Value"]))
lines = []
for k in sorted(agg):
    f, w = statistics.median(agg[k]["FETCH_SIZE"]), statistics.median(agg[k]["WRITE_SIZE"])
    lines.append(f"{k[:70]:70s} fetch raw {f / 1024:8.1f} MiB  write {w / 1024:8.1f} MiB  corrected (2 x FETCH + WRITE) {(2 * f + w) * 1024 / 1e6:8.1f} MB per launch")
open(out + "/TRAFFIC.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
