"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs of the same command, csv output) into HBM bytes per
launch for the convolution-engine kernels (the `roofline.traffic` figure of bench.py).

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> > profiles/<name>.json

Units / corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the
bytes of a wide coalesced read stream (TCC_EA0_RDREQ tallied at 64 B for 128-B requests), so the read side is doubled."""
import csv
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CONV = re.compile(r"sg_conv_v2_kernel|sg_conv_v3_kernel|sg_conv_v4_kernel|sg_conv_sk_kernel|sg_conv_rs_kernel|sg_conv_rs96_kernel|sg_conv_q_kernel|sg_wgrad_q_kernel|sg_wgrad_ql_kernel|k_quad_reduce_fold|"
                  r"sg_wgrad_v2_kernel|sg_wgrad_v3_kernel|sg_wgrad_v3l_kernel|sg_wgrad_sk_kernel|k_splitk_reduce|sg_gemm_kernel<.*ConvPix")


def collect(path, counter):
    per = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            k = re.sub(r"\(.*$", "", r["Kernel_Name"])
            a = per.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return per


def csrc_sha():
    """hash of the kernel sources the traced library was built from (bench.py compares it with its own: a summary of other sources is reported as stale)"""
    import bench
    return bench.csrc_sha16()


def main():
    fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    rows, tot_n, tot_b = [], 0, 0.0
    for k in sorted(set(fetch) | set(write)):
        if not CONV.search(k):
            continue
        n = max(fetch.get(k, [0, 0])[0], write.get(k, [0, 0])[0])
        rd = 2.0 * fetch.get(k, [0, 0.0])[1] * 1024.0
        wr = write.get(k, [0, 0.0])[1] * 1024.0
        rows.append({"kernel": k[:160], "launches": n, "read_GB": round(rd / 1e9, 3), "write_GB": round(wr / 1e9, 3),
                     "bytes_per_launch": round((rd + wr) / max(n, 1))})
        if "k_splitk_reduce" not in k and "k_quad_reduce_fold" not in k:      # the reduce belongs to the weight-gradient launch that precedes it
            tot_n += n
        tot_b += rd + wr
    print(json.dumps({"kernel_family": "convolution engine (sg_conv_v4 / sg_conv_v3 / sg_conv_v2 / sg_conv_sk / sg_conv_rs / sg_wgrad_v3 / sg_wgrad_v2 / sg_wgrad_sk / sg_gemm_kernel<ConvPix*>)", "launches": tot_n,
                      "hbm_bytes_per_launch": round(tot_b / max(tot_n, 1)), "read_side_doubled": True,
                      "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over bench.py --steps 2 --warmup 1 --no-extras --fid-samples 0 --no-cpu-baseline; launches = sg_conv2d_fwd / sg_conv2d_wgrad calls (a weight-gradient launch includes its split-K reduce)",
                      "csrc_sha16": csrc_sha(), "per_kernel": rows}, indent=1))


if __name__ == "__main__":
    main()
