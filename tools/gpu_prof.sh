#!/bin/bash
# GPU box: rocprofv3 kernel trace + stats of the default bench command -> gpurun_out/prof_<tag>/ ; prints the top of the stats table
tag=${1:-b32}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 ${T_PROF:-500} rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --steps ${STEPS:-5} --warmup 2 --no-cpu-baseline --no-secondary "$@" > gpurun_out/prof_$tag.json 2> gpurun_out/prof_$tag.err
echo "rocprof exit $?"
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/prof_${tag}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
fam = {}
for r in rows:
    n = r["Name"]
    m = re.search(r"tav::(\w+)", n)
    k = m.group(1) if m else ("ATen/other: " + n[:60])
    d = fam.setdefault(k, [0, 0.0])
    d[0] += int(r["Calls"]); d[1] += float(r["TotalDurationNs"])
print(f"total kernel time {tot/1e6:.1f} ms over the trace")
for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"{t/tot*100:6.2f} %  {t/1e6:9.2f} ms  calls {c:6d}  avg {t/c/1e3:8.1f} us  {k}")
PY
find gpurun_out/prof_$tag -type f ! -name "*kernel_stats.csv" -delete 2>/dev/null
cat gpurun_out/prof_$tag.json | cut -c1-600
