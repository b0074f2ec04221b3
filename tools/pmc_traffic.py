"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py into profiles/<round>_pmc/traffic_per_launch.json.

On the GPU box (separate passes, counters only with --kernel-trace, as the MI355X guide prescribes):
    cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary
    python3 tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/traffic_per_launch.json --batch 32 --preset B
Per kernel family: KiB per launch averaged over every launch in the trace; corrected HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950
tallies 64 B per 128-B read request, MI355X_MICROARCH.md §HBM; WRITE_SIZE reads exactly for 16-B-per-lane stores).  The file records the
hash of the GEMM kernel sources so that bench.py uses the figure only for the code it was measured on."""
import argparse
import csv
import glob
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csv.field_size_limit(1 << 30)


def family(name):
    m = re.search(r"tav::(\w+)", name)
    return m.group(1) if m else None


def read(dirname, counter, steps_in_trace=1, skip_steps=0):
    """Sum of `counter` and number of launches per kernel family.  The trace holds `steps_in_trace` identical steps (same kernels, same order);
    the first `skip_steps` of them -- the EAGER warm-up before the hipGraph capture -- are dropped per family (by dispatch order), so the
    averages are over replayed steps only."""
    rows = {}
    files = glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        sys.exit(f"no *counter_collection.csv under {dirname}")
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                fam = family(row["Kernel_Name"])
                if fam is None:
                    continue
                rows.setdefault(fam, []).append((int(row.get("Dispatch_Id", 0) or 0), float(row["Counter_Value"])))
    tot, cnt = {}, {}
    for fam, rs in rows.items():
        rs.sort()
        drop = len(rs) * skip_steps // steps_in_trace if (steps_in_trace > 0 and len(rs) % steps_in_trace == 0) else 0
        keep = rs[drop:]
        tot[fam] = sum(v for _, v in keep)
        cnt[fam] = len(keep)
    return tot, cnt


def source_hash(files=("gemm.hip", "common.h")):
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(ROOT, "multi-modal-emotion_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


FAMILY_SOURCES = {"attention": ("attention.hip", "common.h"), "layernorm": ("norm.hip", "common.h"), "gemm_tn": ("gemm.hip", "common.h")}      # as in bench.py


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir")
    ap.add_argument("write_dir")
    ap.add_argument("out")
    ap.add_argument("--batch", type=int, required=True)
    ap.add_argument("--preset", default="B")
    ap.add_argument("--head", default="")
    ap.add_argument("--steps-in-trace", type=int, default=5, help="steps the profiled command executed (warm-up + 2 post-capture replays + timed steps)")
    ap.add_argument("--skip-steps", type=int, default=1, help="leading EAGER steps to drop (the warm-up before the capture)")
    a = ap.parse_args()
    ft, fc = read(a.fetch_dir, "FETCH_SIZE", a.steps_in_trace, a.skip_steps)
    wt, wc = read(a.write_dir, "WRITE_SIZE", a.steps_in_trace, a.skip_steps)
    fams = {}
    for k in sorted(ft):
        if k not in wt:
            continue
        f_kib, w_kib = ft[k] / fc[k], wt[k] / wc[k]
        fams[k] = {"launches_in_trace": fc[k], "fetch_kib_raw_per_launch": round(f_kib, 1), "write_kib_per_launch": round(w_kib, 1),
                   "hbm_bytes_per_launch_corrected": int((2 * f_kib + w_kib) * 1024)}
    head = a.head
    if not head:
        try:
            head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
        except Exception:
            head = ""
    rec = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, with --kernel-trace only) over `python3 bench.py --steps 2 --warmup 1 "
                   "--no-cpu-baseline --no-roofline --no-secondary`; the eager warm-up step is dropped (first 1/5 of every family's dispatches): KiB per launch "
                   "averaged over the REPLAYED steps only; raw per-dispatch CSVs beside this file (*.csv.gz); corrected = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 counts 64 B per 128-B read request, MI355X_MICROARCH.md)",
           "kernel_source_hash": source_hash(), "family_source_hashes": {k: source_hash(v) for k, v in FAMILY_SOURCES.items()}, "head": head, "preset": a.preset, "per_gpu_batch": a.batch, "families": fams}
    with open(a.out, "w") as f:
        json.dump(rec, f, indent=1)
    for k, v in fams.items():
        print(f"{k:28s} launches {v['launches_in_trace']:6d}  fetch raw {v['fetch_kib_raw_per_launch'] / 1024:9.1f} MiB  write {v['write_kib_per_launch'] / 1024:9.1f} MiB  "
              f"corrected {v['hbm_bytes_per_launch_corrected'] / 1e6:9.1f} MB")


if __name__ == "__main__":
    main()
