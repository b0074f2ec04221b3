"""Which kernels stand around the host framework's fill / copy launches?  Reads a rocprofv3 --kernel-trace CSV of `bench.py --profile-serial`
(one stream, eager: dispatch order = program order) and prints, for every kernel whose name matches the pattern, the most common
(previous kernel, grid size, next kernel) triples.     usage: python tools/trace_neighbours.py <kernel_trace.csv> [pattern]"""
import collections
import csv
import re
import sys

csv.field_size_limit(1 << 30)
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else r"FillFunctor|copyBuffer")


def short(n):
    m = re.search(r"tav::(\w+)", n)
    if m:
        return m.group(1)
    m = re.search(r"at::native::(?:\(anonymous namespace\)::)?(\w+)<[^,]*,?\s*(?:at::native::)?(\w+)?", n)
    return ("aten:" + m.group(1) + ":" + (m.group(2) or "")) if m else n[:50]


rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
cnt = collections.Counter()
for i, n in enumerate(names):
    if pat.search(n):
        prev = short(names[i - 1]) if i else "-"
        nxt = short(names[i + 1]) if i + 1 < len(names) else "-"
        cnt[(short(n), rows[i].get("Grid_Size", "?"), prev, nxt)] += 1
print(f"{sum(cnt.values())} matching launches among {len(names)}")
for (n, grid, prev, nxt), c in cnt.most_common(40):
    print(f"{c:6d}  {n:40s} grid {grid:>10s}   after {prev:32s} before {nxt}")
