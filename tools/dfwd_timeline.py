"""Per-launch timeline of the LAST discriminator forward in a rocprofv3 kernel trace of bench.py (its D-forward leg runs last):
start offset, duration and kernel of every dispatch from the NCHW->NHWC conversion of the input image to the projection head, plus
the sum over the convolution-engine launches (the `d_forward_stack.conv_stack_ms` figure of bench.py, per launch).

    rocprofv3 --kernel-trace --output-format csv -d out -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --fid-samples 0
    python tools/dfwd_timeline.py out/*/*_kernel_trace.csv
"""
import csv
import re
import sys

CONV = re.compile(r"sg_conv_v2_kernel|sg_conv_v3_kernel|sg_conv_v4_kernel|sg_conv_sk_kernel|sg_conv_rs_kernel|sg_conv_rs96_kernel|sg_conv_q_kernel|sg_gemm_kernel<.*ConvPix")


def short(n):
    n = re.sub(r"\(.*$", "", n).replace("unsigned short", "bf16")
    return n[:110]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if "k_nchw_to_nhwc" in r["Kernel_Name"]]
    if not starts:
        sys.exit("no k_nchw_to_nhwc dispatch in the trace (not a bench.py trace?)")
    last = rows[starts[-1]:]
    end = next((i for i, r in enumerate(last) if "k_pd_head_fwd" in r["Kernel_Name"]), len(last) - 1)
    last = last[:end + 1]
    t0 = int(last[0]["Start_Timestamp"])
    conv_us, n_conv = 0.0, 0
    print(f"{'start us':>9} {'dur us':>8}  kernel")
    for r in last:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = short(r["Kernel_Name"])
        if CONV.search(name):
            conv_us += (e - s) / 1e3
            n_conv += 1
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {name}")
    span = (int(last[-1]["End_Timestamp"]) - t0) / 1e3
    print(f"forward span {span / 1e3:.3f} ms; {n_conv} convolution-engine launches, {conv_us / 1e3:.3f} ms")


if __name__ == "__main__":
    main()
