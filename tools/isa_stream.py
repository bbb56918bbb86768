"""Static look at the NON-MFMA (streaming) kernels of libsgamd.so, the companion of tools/isa_mix.py: per kernel, the vector-ALU cost of its main loop per
kilobyte of global traffic the loop issues (one 16-byte-per-lane instruction = 1 KB per wave). The HBM delivers ~13 B / clk / CU at 8 TB/s, i.e. one KB
per ~310 clk per SIMD: a loop that spends more vector-pipe clocks than that per KB it moves (full-rate instruction = 4 clk per wave64, 32-bit integer
multiply / transcendental = 16) cannot stream at the roof whatever its memory pattern is. Counts are static (every instruction of the largest backward-
branch range once; the whole body when there is no loop): an indicator of where to look, not a time.

    python tools/isa_stream.py [--so pytorch-studiogan_amd/libsgamd.so] [--match k_bn,k_sn]
"""
import argparse
import os
import re
import subprocess
import tempfile

import isa_mix

LLVM = isa_mix.LLVM
VMEM_BYTES = {"dwordx4": 16, "dwordx3": 12, "dwordx2": 8, "dword": 4, "ushort": 2, "short": 2, "ubyte": 1, "byte": 1, "sbyte": 1, "sshort": 2, "short_d16": 2, "short_d16_hi": 2}
SLOW = isa_mix.QUARTER_RATE + ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64",
                              "v_fma_f64", "v_mul_f64", "v_add_f64", "v_div_scale_f64", "v_div_fmas_f64", "v_div_fixup_f64")


def main():
    ap = argparse.ArgumentParser()
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ap.add_argument("--so", default=os.path.join(here, "pytorch-studiogan_amd", "libsgamd.so"))
    ap.add_argument("--match", default="")
    args = ap.parse_args()
    rows = []
    with tempfile.TemporaryDirectory() as td:
        for k, co in enumerate(isa_mix.code_objects(args.so)):
            f = os.path.join(td, f"co{k}.o")
            open(f, "wb").write(co)
            dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", f], capture_output=True, text=True).stdout
            cur, body = None, {}
            for line in dis.split("\n"):
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    cur = m.group(1)
                    body[cur] = []
                    continue
                if cur is None or not line.strip():
                    continue
                a = re.search(r"// ([0-9A-F]+):", line)
                body[cur].append((int(a.group(1), 16) if a else -1, line))
            for name, ins in body.items():
                if not ins or any(l.split()[0].startswith("v_mfma") for _, l in ins):
                    continue
                base = ins[0][0]
                best = None
                for addr, line in ins:
                    t = line.split()
                    if t and t[0].startswith("s_cbranch"):
                        m = re.search(r"\+0x([0-9a-f]+)>", line)
                        tgt = base + int(m.group(1), 16) if m else None
                        if tgt is not None and tgt < addr:
                            nv = sum(1 for a2, l2 in ins if tgt <= a2 <= addr and l2.split()[0].startswith(("global_", "buffer_", "flat_")))
                            if nv and (best is None or addr - tgt > best[1] - best[0]):
                                best = (tgt, addr)
                d = dict(valu=0, slow=0, ld=0, st=0, ds=0, n=0, loop=best is not None)
                for addr, line in ins:
                    if best is not None and not (best[0] <= addr <= best[1]):
                        continue
                    op = line.split()[0]
                    d["n"] += 1
                    if op.startswith(("global_", "buffer_", "flat_")):
                        w = next((b for k2, b in VMEM_BYTES.items() if op.endswith("_" + k2)), 4)
                        if "atomic" in op:
                            d["st"] += 4 * 64
                        elif "store" in op:
                            d["st"] += w * 64
                        else:
                            d["ld"] += w * 64
                    elif op.startswith("ds_"):
                        d["ds"] += 1
                    elif op.startswith("v_"):
                        d["valu"] += 1
                        if op.startswith(SLOW):
                            d["slow"] += 1
                rows.append((name, d))
    names = isa_mix.demangle([r[0] for r in rows])
    sel = [s for s in args.match.split(",") if s]
    print(f"{'instr':>5s} {'VALU':>5s} {'slow':>4s} {'LDS':>4s} {'ld B':>6s} {'st B':>6s} {'clk/KB':>7s} | kernel (clk/KB = vector-pipe clocks per KB of global traffic of the loop; > ~310 cannot reach 8 TB/s; * = no loop with global traffic: whole body)")
    for (raw, d), nm in sorted(zip(rows, names), key=lambda t: t[1]):
        if sel and not any(s in nm for s in sel):
            continue
        kb = (d["ld"] + d["st"]) / 1024.0
        if kb == 0:
            continue
        clk = (4 * (d["valu"] - d["slow"]) + 16 * d["slow"]) / kb
        nm = re.sub(r"\(.*$", "", nm)
        print(f"{d['n']:5d} {d['valu']:5d} {d['slow']:4d} {d['ds']:4d} {d['ld']:6d} {d['st']:6d} {clk:7.0f} | {nm}{'' if d['loop'] else ' *'}")


if __name__ == "__main__":
    main()
