"""Cross-check bench.py's live `roofline` figures (hipEvent brackets, csrc/capi.hip) against the rocprofv3 kernel trace of the same
process: sums the convolution-engine kernels over the timed steps of the trace and prints both side by side.

    rocprofv3 --kernel-trace --output-format csv -d out -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --fid-samples 0 > bench.log
    python tools/roofline_crosscheck.py out/*/*_kernel_trace.csv bench.log 1 3 > profiles/<name>.txt        (warmup steps, timed steps)
"""
import csv
import json
import re
import sys

CONV = re.compile(r"sg_conv_v2_kernel|sg_conv_v3_kernel|sg_conv_v4_kernel|sg_conv_sk_kernel|sg_conv_rs_kernel|sg_wgrad_v2_kernel|sg_wgrad_v3_kernel|sg_wgrad_sk_kernel|k_splitk_reduce|sg_gemm_kernel<.*ConvPix")


def main():
    trace, log, warm, steps = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    rows = list(csv.DictReader(open(trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    adam = [i for i, r in enumerate(rows) if "k_adam_ema" in r["Kernel_Name"]]      # 3 optimizer launches per step (2 D updates + 1 G update)
    last = [adam[3 * k + 2] for k in range(len(adam) // 3)]
    lo, hi = last[warm - 1] + 1, last[warm + steps - 1] + 1
    sel = [r for r in rows[lo:hi] if CONV.search(r["Kernel_Name"])]
    tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel) / 1e6
    kern = sum(1 for r in sel if "k_splitk_reduce" not in r["Kernel_Name"])
    r = json.loads([l for l in open(log) if l.startswith("{")][0])["roofline"]
    print(f"rocprofv3: {len(sel)} conv-engine dispatches in the {steps} timed steps ({kern} contraction kernels + {len(sel) - kern} split-K "
          f"reductions), {tot:.2f} ms of kernel time = {tot / steps:.2f} ms per step, {tot / kern * 1e3:.1f} us per contraction launch")
    print(f"bench.py : launches_per_step {r['launches_per_step']}, conv_ms_per_step {r['conv_ms_per_step']}, avg_launch_ms {r['avg_launch_ms']}, "
          f"achieved {r['achieved']} TFLOP/s (frac {r['frac']})")


if __name__ == "__main__":
    main()
