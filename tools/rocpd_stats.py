"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / share, like --stats.
usage: python tools/rocpd_stats.py gpurun_out/prof/x_results.db [--top N] > profiles/<name>.txt"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    m = re.match(r"void sg_gemm_kernel<(.*)", name)
    if m:
        a = m.group(1)
        a = a.replace("unsigned short", "bf16").replace("float", "f32")
        a = re.sub(r"(StridedKC|StridedMC|ConvPixKC|ConvPixMC)<[^>]*>", r"\1", a)
        return "sg_gemm_kernel<" + a
    return name[:110]


def csv_rows(path):
    import csv
    agg = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            n = r["Kernel_Name"]
            t = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            a = agg.setdefault(n, [0, 0, 1 << 62, 0])
            a[0] += 1; a[1] += t; a[2] = min(a[2], t); a[3] = max(a[3], t)
    return [(short(n), a[0], a[1], a[2], a[3]) for n, a in agg.items()]


def main():
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
    if sys.argv[1].endswith(".csv"):
        rows = csv_rows(sys.argv[1])
    else:
        db = sqlite3.connect(sys.argv[1])
        q = """select s.kernel_name, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start)
               from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name"""
        rows = [(short(n), c, t, mn, mx) for n, c, t, mn, mx in db.execute(q)]
    agg = {}
    for n, c, t, mn, mx in rows:
        a = agg.setdefault(n, [0, 0, 1 << 62, 0])
        a[0] += c; a[1] += t; a[2] = min(a[2], mn); a[3] = max(a[3], mx)
    total = sum(a[1] for a in agg.values())
    print(f"total kernel time {total / 1e6:.2f} ms over {sum(a[0] for a in agg.values())} dispatches")
    print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>10} {'share':>7}  kernel")
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{a[0]:7d} {a[1] / 1e6:10.2f} {a[1] / a[0] / 1e3:10.1f} {a[2] / 1e3:9.1f} {a[3] / 1e3:10.1f} {100.0 * a[1] / total:6.1f}%  {n}")


if __name__ == "__main__":
    main()
