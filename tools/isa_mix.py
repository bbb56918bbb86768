"""Static look at the gfx950 code objects inside libsgamd.so (no GPU needed): registers, scratch, static LDS (the convolution kernels size theirs at launch: 0 here), waves per SIMD the register count allows,
and the instruction mix of every kernel that issues MFMAs -- LDS bytes read per MFMA, vector-ALU instructions per MFMA, full LDS waits
(`s_waitcnt lgkmcnt(0)`) per MFMA. Counts are STATIC and taken over the kernel's MAIN LOOP = the largest backward-branch range that contains
MFMAs (every instruction of that range once, inner loops not weighted, both sides of a branch counted; the whole body when the kernel has no
such loop): they say nothing about time -- the use is to spot spills and to compare the operand traffic of two kernels for the same MFMA
(one 32x32x16 bf16 MFMA = 32 cycles of one matrix pipe; the LDS delivers 128 B / clk / CU, i.e. 1 KB per MFMA slot with all four pipes busy).

    python tools/isa_mix.py [--so pytorch-studiogan_amd/libsgamd.so] [--match conv_q,wgrad] [--all]
"""
import argparse
import os
import re
import struct
import subprocess
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
DS_BYTES = {"ds_read_b128": 16, "ds_read_b64": 8, "ds_read_b32": 4, "ds_read2_b64": 16, "ds_read2_b32": 8, "ds_read_b64_tr_b16": 8,
            "ds_read_b96": 12, "ds_read_u16": 2, "ds_read_u8": 1, "ds_read2st64_b64": 16, "ds_read2st64_b32": 8, "ds_read_b64_tr_b8": 8}


# 32-bit integer multiplies issue at a quarter of the vector-ALU rate (16 clk per wave64 instead of 4)
QUARTER_RATE = ("v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32", "v_mul_lo_i32", "v_mad_u64_u32", "v_mad_i64_i32")


def code_objects(so):
    """the gfx950 members of every clang offload bundle embedded in the shared library"""
    data = open(so, "rb").read()
    out = []
    for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data):
        p = m.start()
        n = struct.unpack_from("<Q", data, p + 24)[0]
        off = p + 32
        for _ in range(n):
            o, sz, tl = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + tl].decode()
            off += tl
            if "gfx950" in triple and sz:
                out.append(data[p + o:p + o + sz])
    return out


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return r.stdout.split("\n")[:len(names)]


def main():
    ap = argparse.ArgumentParser()
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ap.add_argument("--so", default=os.path.join(here, "pytorch-studiogan_amd", "libsgamd.so"))
    ap.add_argument("--match", default="", help="comma-separated substrings of the demangled name")
    ap.add_argument("--all", action="store_true", help="kernels without MFMAs too")
    args = ap.parse_args()
    rows = []
    with tempfile.TemporaryDirectory() as td:
        for k, co in enumerate(code_objects(args.so)):
            f = os.path.join(td, f"co{k}.o")
            open(f, "wb").write(co)
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", f], capture_output=True, text=True).stdout
            meta = {}
            for e in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
                g = lambda key: (re.search(r"\." + key + r":\s+(\S+)", e) or [None, "0"])[1]
                meta[g("name")] = dict(agpr=int(e.split()[0]), vgpr=int(g("vgpr_count")), sgpr=int(g("sgpr_count")), scratch=int(g("private_segment_fixed_size")),
                                       lds=int(g("group_segment_fixed_size")), spill=int(g("vgpr_spill_count")))
            dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", f], capture_output=True, text=True).stdout
            body, cur = {}, None
            for line in dis.split("\n"):
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    cur = m.group(1) if m.group(1) in meta else None
                    if cur:
                        body[cur] = []
                    continue
                if cur is None or not line.strip():
                    continue
                a = re.search(r"// ([0-9A-F]+):", line)
                body[cur].append((int(a.group(1), 16) if a else -1, line))
            for name, ins in body.items():
                # main loop: the largest [target, branch] range of a backward branch that holds MFMAs
                base = ins[0][0] if ins else 0
                best = None
                for addr, line in ins:
                    t = line.split()
                    if t and t[0].startswith("s_cbranch"):
                        m = re.search(r"\+0x([0-9a-f]+)>", line)
                        tgt = base + int(m.group(1), 16) if m else (base if re.search(r"<[^+>]+>\s*$", line) else None)
                        if tgt is not None and tgt < addr:
                            nm = sum(1 for a2, l2 in ins if tgt <= a2 <= addr and l2.split()[0].startswith(("v_mfma", "v_smfmac")))
                            if nm and (best is None or addr - tgt > best[1] - best[0]):
                                best = (tgt, addr)
                d = meta[name]
                d.update(mfma=0, ds=0, dsn=0, valu=0, qmul=0, salu=0, vmem=0, ldsdma=0, wait0=0, waits=0, barrier=0, n=0, loop=best is not None)
                for addr, line in ins:
                    if best is not None and not (best[0] <= addr <= best[1]):
                        continue
                    op = line.split()[0]
                    d["n"] += 1
                    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
                        d["mfma"] += 1
                    elif op.startswith("ds_read"):
                        d["ds"] += DS_BYTES.get(op, 8) * 64
                        d["dsn"] += 1
                    elif op.startswith(("buffer_", "global_", "scratch_", "flat_")):
                        d["vmem"] += 1
                        if " lds" in line:
                            d["ldsdma"] += 1
                    elif op.startswith("v_"):
                        d["valu"] += 1
                        if op.startswith(QUARTER_RATE):
                            d["qmul"] += 1
                    elif op == "s_waitcnt":
                        d["waits"] += 1
                        if "lgkmcnt(0)" in line:
                            d["wait0"] += 1
                    elif op == "s_barrier":
                        d["barrier"] += 1
                    elif op.startswith("s_"):
                        d["salu"] += 1
            for name, d in meta.items():
                if "n" in d:
                    rows.append((name, d))
    names = demangle([r[0] for r in rows])
    sel = [s for s in args.match.split(",") if s]
    print(f"{'vgpr':>4s} {'agpr':>4s} {'w/SIMD':>6s} {'sLDS K':>6s} {'scr B':>5s} {'spill':>5s} | {'MFMA':>5s} {'dsB/MFMA':>8s} {'VALU/MFMA':>9s} {'qmul':>4s} {'lgkm0/MFMA':>10s} {'LDS-DMA':>7s} {'barr':>4s} | kernel (* = no MFMA loop found: whole body)")
    for (raw, d), nm in sorted(zip(rows, names), key=lambda t: t[1]):
        if sel and not any(s in nm for s in sel):
            continue
        if not d["mfma"] and not args.all and not d["scratch"]:
            continue
        regs = (d["vgpr"] + 7) // 8 * 8          # .vgpr_count is the unified total (architectural + accumulation registers)
        occ = min(8, 512 // max(regs, 1))
        mf = max(d["mfma"], 1)
        nm = re.sub(r"\(.*$", "", nm)
        print(f"{d['vgpr']:4d} {d['agpr']:4d} {occ:6d} {d['lds'] / 1024:6.1f} {d['scratch']:5d} {d['spill']:5d} | {d['mfma']:5d} {d['ds'] / mf:8.0f} {d['valu'] / mf:9.1f} {d['qmul']:4d} "
              f"{d['wait0'] / mf:10.2f} {d['ldsdma']:7d} {d['barrier']:4d} | {nm}{'' if d['loop'] else ' *'}")


if __name__ == "__main__":
    main()
