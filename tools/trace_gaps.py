"""Kernel-trace analysis (rocprofv3 --kernel-trace CSV): over the LAST replayed step(s) of the trace, how long is the device busy with at least one
kernel, how long idle, how many kernels overlap on average, and per queue: busy time, kernels, gaps between consecutive kernels."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]) for r in rows]
ev.sort()
# the last 3 steps: find the AdamW kernels as step delimiters
adam = [e for e in ev if "adamw_chunk_kernel" in e[3]]
if len(adam) < 4:
    print("not enough steps in the trace"); sys.exit(0)
t0, t1 = adam[-4][1], adam[-1][1]
nsteps = 3
win = [e for e in ev if e[0] >= t0 and e[1] <= t1]
print(f"window: {nsteps} steps, {(t1 - t0) / nsteps / 1e6:.3f} ms per step, {len(win) / nsteps:.0f} kernels per step")
# union busy time
pts = []
for s, e, _, _ in win:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
busy = 0; depth = 0; last = None; area = 0
for t, d in pts:
    if depth > 0:
        busy += t - last; area += (t - last) * depth
    depth += d; last = t
print(f"device busy (>= 1 kernel): {busy / nsteps / 1e6:.3f} ms per step; idle {(t1 - t0 - busy) / nsteps / 1e6:.3f} ms per step; mean kernels in flight while busy {area / max(busy, 1):.2f}")
print(f"sum of kernel durations: {sum(e - s for s, e, _, _ in win) / nsteps / 1e6:.3f} ms per step")
byq = collections.defaultdict(list)
for s, e, q, n in win:
    byq[q].append((s, e, n))
for q, lst in sorted(byq.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
    lst.sort()
    dur = sum(e - s for s, e, _ in lst)
    gaps = [lst[i + 1][0] - lst[i][1] for i in range(len(lst) - 1)]
    small = [g for g in gaps if 0 <= g < 20000]
    print(f"queue {q}: {len(lst) / nsteps:.0f} kernels/step, busy {dur / nsteps / 1e6:.3f} ms/step, gaps < 20 us: {len(small) / nsteps:.0f}/step totalling {sum(small) / nsteps / 1e6:.3f} ms/step"
          f" (median {sorted(small)[len(small) // 2] / 1e3 if small else 0:.2f} us)")
