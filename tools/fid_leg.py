"""The FID feature-extraction leg alone (reference src/metrics/features.py:17-65): G_ema forward (bf16) + on-device quantise / resize +
InceptionV3 + moments, for a kernel trace that is not mixed with the training step:
    cd /tmp && rocprofv3 --kernel-trace --stats -d <out> -o kt --output-format csv -- python <repo>/tools/fid_leg.py --samples 5120 --dtype bf16
Prints one JSON line: samples/s and the leg's algorithmic roofline (G forward 42.24 GFLOP + InceptionV3 11.4 GFLOP per sample)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

G_FWD_GFLOP = 42.24          # SURVEY Appendix A.2 (BigGAN-128 generator forward, per image)
INCEPTION_GFLOP = 11.4       # InceptionV3 at 299 x 299, 2 * MACs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=5120)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--f32-mode", default="exact", choices=["exact", "bf16x3"], help="fp32 Inception: exact fp32 MFMA or the bf16x3 split (functional.f32_mode)")
    args = ap.parse_args()
    from studiogan_amd import metrics as M
    dev = torch.device("cuda:0")
    wl = bench.WORKLOADS["biggan128"]
    G, _ = bench.build(wl, True, dev)
    G.eval()
    for p in G.parameters():
        p.requires_grad_(False)
    idt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    model = M.LoadEvalModel(device=dev, state_dict=M.synthetic_state_dict(0), dtype=idt, f32_mode=args.f32_mode)
    M.generate_images_and_stack_features(G, model, 2 * args.batch, args.batch, wl["z_dim"], wl["classes"], device=dev)
    torch.cuda.synchronize()
    mom = M.FeatureMoments(2048, dev)
    t0 = time.perf_counter()
    M.generate_images_and_stack_features(G, model, args.samples, args.batch, wl["z_dim"], wl["classes"], quantize=True, device=dev, moments=mom)
    mom.finalize(None)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sps = args.samples / dt
    tf = sps * (G_FWD_GFLOP + INCEPTION_GFLOP) / 1e3
    print(json.dumps({"metric": "FID feature-extract samples/sec", "value": round(sps, 1), "samples": args.samples, "inception_dtype": args.dtype, "f32_mode": args.f32_mode if args.dtype == "f32" else None,
                      "roofline": {"bound": "mfma", "achieved": round(tf, 1), "unit": "TFLOP/s", "peak": 2500.0, "frac": round(tf / 2500.0, 4),
                                   "gflop_per_sample": {"generator_forward_bf16": G_FWD_GFLOP, "inception_v3": INCEPTION_GFLOP},
                                   "note": "fp32 Inception runs on the exact-fp32 MFMA path (157 TFLOP/s peak): its frac is against the bf16 peak only for comparability"}}))


if __name__ == "__main__":
    main()
