"""Per-kernel summary of rocprofv3 --pmc counter_collection CSVs (one or more passes of the same command):
    python tools/pmc_summary.py gpurun_out/r2a/pmc1/pmc1_counter_collection.csv [more.csv ...] > profiles/r02_conv_sq_counters.txt
For every kernel family: dispatches, mean duration, and the per-dispatch mean of every counter; derived columns:
  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs * 256 CUs * duration * clock)   (busy cycles are summed over SIMDs; clock from SQ_BUSY_CYCLES / duration when available)
  lds_stall = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES, wait_any = SQ_WAIT_ANY / SQ_WAVE_CYCLES, issue_stall = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (quad-cycle units both)
  bank_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "")
    return name[:90]


def main():
    per = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    dur = defaultdict(lambda: [0, 0.0])
    seen = set()
    for path in sys.argv[1:]:
        with open(path) as f:
            for r in csv.DictReader(f):
                k = short(r["Kernel_Name"])
                if not k.startswith(("sg_", "k_")):
                    continue
                c = per[k][r["Counter_Name"]]
                c[0] += 1
                c[1] += float(r["Counter_Value"])
                key = (path, r["Dispatch_Id"])
                if key not in seen:
                    seen.add(key)
                    d = dur[k]
                    d[0] += 1
                    d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    names = sorted({c for k in per for c in per[k]})
    print("# per-dispatch means; durations in us (profiled passes clock lower than un-profiled runs: compare ratios, not absolute times)")
    print(f"{'kernel':92s} {'n':>5s} {'us':>9s} " + " ".join(f"{n[:22]:>22s}" for n in names) + "   derived")
    for k in sorted(per, key=lambda k: -dur[k][1]):
        n, tot = dur[k]
        us = tot / max(n, 1)
        m = {c: per[k][c][1] / max(per[k][c][0], 1) for c in per[k]}
        der = []
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and us > 0:
            clk = 2.0e3  # cycles per us at ~2.0 GHz under load
            der.append(f"mfma_busy={m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * us * clk):.3f}")
        if "SQ_WAVE_CYCLES" in m and m["SQ_WAVE_CYCLES"] > 0:
            for a, b in (("SQ_WAIT_INST_LDS", "lds_stall"), ("SQ_WAIT_ANY", "wait_any"), ("SQ_WAIT_INST_ANY", "issue_stall"), ("SQ_ACTIVE_INST_ANY", "active")):
                if a in m:
                    der.append(f"{b}={m[a] / m['SQ_WAVE_CYCLES']:.3f}")
        if "SQ_LDS_BANK_CONFLICT" in m and m.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
            der.append(f"bank_conflict={m['SQ_LDS_BANK_CONFLICT'] / m['SQ_LDS_IDX_ACTIVE']:.3f}")
        print(f"{k:92s} {n:5d} {us:9.1f} " + " ".join(f"{m.get(c, float('nan')):22.4g}" for c in names) + "   " + " ".join(der))


if __name__ == "__main__":
    main()
