"""Summarise the per-dispatch counter CSVs of tools/gpu_pmc_attn.sh: median per kernel and counter, and the derived shares."""
import collections
import csv
import glob
import os
import statistics
import sys

out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in sorted(glob.glob(out + "/pass*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if os.environ.get("TAV_PMC_FILTER", "attn_") not in k:
            continue
        k = k.split("(")[0].replace("void tav::", "")
        if "Grid_Size" in r and "gemm" in k:
            k += f" grid {r['Grid_Size']}"
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if "Start_Timestamp" in r and r["Counter_Name"] in ("SQ_WAVE_CYCLES", "SQ_INSTS_VALU"):
            dur[k].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
lines = []
for k in sorted(agg):
    m = {c: statistics.median(v) for c, v in agg[k].items()}
    wc = m.get("SQ_WAVE_CYCLES", 0.0)
    lines.append(f"## {k}   (median over {len(next(iter(agg[k].values())))} dispatches; duration under the profiler {statistics.median(dur[k]) if dur[k] else 0:.1f} us)")
    lines.append("   " + "  ".join(f"{c}={v:.4g}" for c, v in sorted(m.items())))
    if wc:
        sh = lambda c: 100.0 * m.get(c, 0.0) / wc
        lines.append(f"   shares of SQ_WAVE_CYCLES: wait(waitcnt/barrier) {sh('SQ_WAIT_ANY'):.1f} %  issue-stall {sh('SQ_WAIT_INST_ANY'):.1f} %  active {sh('SQ_ACTIVE_INST_ANY'):.1f} %"
                     f"  [VALU {sh('SQ_ACTIVE_INST_VALU'):.1f}  LDS {sh('SQ_ACTIVE_INST_LDS'):.1f}  VMEM {sh('SQ_ACTIVE_INST_VMEM'):.1f}  SCA {sh('SQ_ACTIVE_INST_SCA'):.1f}  MISC {sh('SQ_ACTIVE_INST_MISC'):.1f}]  LDS-issue-stall {sh('SQ_WAIT_INST_LDS'):.1f} %")
    if m.get("SQ_INSTS_MFMA"):
        nm = m["SQ_INSTS_MFMA"]
        lines.append(f"   per MFMA: VALU {m.get('SQ_INSTS_VALU', 0) / nm:.2f} (transcendental {m.get('SQ_INSTS_VALU_TRANS_F32', 0) / nm:.2f}, cvt {m.get('SQ_INSTS_VALU_CVT', 0) / nm:.2f})  SALU {m.get('SQ_INSTS_SALU', 0) / nm:.2f}  LDS {m.get('SQ_INSTS_LDS', 0) / nm:.2f}"
                     f"   MFMA busy cycles / MFMA {m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / nm:.1f}   MFMA-VALU coexec cycles / MFMA busy {m.get('SQ_VALU_MFMA_COEXEC_CYCLES', 0) / max(m.get('SQ_VALU_MFMA_BUSY_CYCLES', 1), 1):.2f}")
    if m.get("SQ_BUSY_CYCLES") and m.get("SQ_VALU_MFMA_BUSY_CYCLES") and m.get("GRBM_GUI_ACTIVE"):
        lines.append(f"   MFMA util = MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 XCD * 1024 SIMDs) = {100.0 * m['SQ_VALU_MFMA_BUSY_CYCLES'] / (m['GRBM_GUI_ACTIVE'] / 8 * 1024):.1f} %   waves {m.get('SQ_WAVES', 0):.0f}")
    if m.get("SQ_LDS_IDX_ACTIVE"):
        lines.append(f"   LDS bank conflict / active = {m.get('SQ_LDS_BANK_CONFLICT', 0) / m['SQ_LDS_IDX_ACTIVE']:.3f}; unaligned stall {m.get('SQ_LDS_UNALIGNED_STALL', 0):.3g}; data-fifo-full {m.get('SQ_LDS_DATA_FIFO_FULL', 0):.3g}")
txt = "\n".join(lines)
print(txt)
open(out + "/SUMMARY.txt", "w").write(txt + "\n")
