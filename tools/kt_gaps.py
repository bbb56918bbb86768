"""Is the traced step GPU-bound? From a rocprofv3 --kernel-trace CSV: the window of the last `steps` training steps (delimited by k_adam_ema launches, 3 per C3 step),
busy time (union of kernel intervals) / span, the idle time by size class, and the kernels that most often sit in front of an idle gap.
    python tools/kt_gaps.py <..._kernel_trace.csv> [steps=6] [adams_per_step=3] [skip_last_steps=2]
(bench.py runs two more steps after the timed ones, with a host synchronisation in front: skip_last_steps keeps them out of the window.)"""
import csv
import re
import sys
from collections import defaultdict


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*$", "", r["Kernel_Name"])[:90]))
    rows.sort()
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    per = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    adams = [i for i, r in enumerate(rows) if r[2].startswith("k_adam_ema")]
    if len(adams) < steps * per + 1:
        print(f"only {len(adams)} k_adam_ema launches in the trace"); return
    skip = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    if len(adams) < (steps + skip) * per + 1:
        print(f"only {len(adams)} k_adam_ema launches in the trace"); return
    i0, i1 = adams[-(steps + skip) * per - 1] + 1, adams[-skip * per - 1] + 1 if skip else adams[-1] + 1
    win = rows[i0:i1]
    span = win[-1][1] - win[0][0]
    busy, cur_end, gaps = 0, win[0][0], []
    for k, (a, b, n) in enumerate(win):
        if a > cur_end:
            gaps.append((a - cur_end, win[k - 1][2], n))
            busy += b - a
            cur_end = b
        elif b > cur_end:
            busy += b - cur_end
            cur_end = b
    print(f"window: {len(win)} dispatches over {steps} steps, span {span / 1e6:.3f} ms = {span / 1e6 / steps:.3f} ms per step; busy {busy / 1e6:.3f} ms = {100.0 * busy / span:.2f} % of the span; "
          f"sum of kernel durations {sum(b - a for a, b, _ in win) / 1e6:.3f} ms")
    idle = span - busy
    print(f"idle {idle / 1e6:.3f} ms = {idle / 1e6 / steps:.3f} ms per step in {len(gaps)} gaps")
    for lo, hi in ((0, 2e3), (2e3, 5e3), (5e3, 10e3), (10e3, 20e3), (20e3, 50e3), (50e3, 1e12)):
        g = [x[0] for x in gaps if lo <= x[0] < hi]
        print(f"  gaps {lo / 1e3:5.0f}..{min(hi, 1e9) / 1e3:7.0f} us: {len(g):6d}  total {sum(g) / 1e6 / steps:7.3f} ms per step")
    by = defaultdict(lambda: [0, 0])
    for g, prev, nxt in gaps:
        by[nxt][0] += 1; by[nxt][1] += g
    print("idle in FRONT of (the kernel that started late), per step:")
    for n, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"  {c / steps:7.1f} gaps {t / 1e6 / steps:7.3f} ms  {n}")


if __name__ == "__main__":
    main()
