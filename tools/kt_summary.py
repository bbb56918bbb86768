"""Per-kernel summary of a rocprofv3 --kernel-trace CSV (same table as tools/rocpd_stats.py makes from a rocpd database):
    python tools/kt_summary.py <..._kernel_trace.csv> [top]"""
import csv
import re
import sys
from collections import defaultdict


def main():
    per = defaultdict(list)
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            n = re.sub(r"\(.*$", "", r["Kernel_Name"])[:110]
            per[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 80
    total = sum(sum(v) for v in per.values())
    print(f"total kernel time {total / 1e3:.2f} ms over {sum(len(v) for v in per.values())} dispatches")
    print(f"{'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>10s} {'share':>7s}  kernel")
    for n, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:top]:
        print(f"{len(v):7d} {sum(v) / 1e3:10.2f} {sum(v) / len(v):10.1f} {min(v):9.1f} {max(v):10.1f} {100 * sum(v) / total:6.1f}%  {n}")


if __name__ == "__main__":
    main()
