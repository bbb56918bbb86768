"""Source-to-source step of hipemu (TEST INFRASTRUCTURE): rewrites the gfx950-only constructs of pytorch-studiogan_amd/csrc/*.h / *.hip so that a HOST
compiler can build the kernels against tests/hipemu/include/hip/hip_runtime.h. Nothing is re-implemented here: the kernel bodies, their index
arithmetic and their launchers are compiled as they stand; only these constructs change form

    asm volatile("s_waitcnt vmcnt(N)" ...)                  -> hipemu::waitcnt_vm(N)           (drains the wave's queued LDS-DMA transfers)
    asm volatile("s_waitcnt lgkmcnt(..)" ...)               -> hipemu::wave_lockstep()          (LDS accesses complete at once in the model, but the lanes of a
                                                               wave are fibres: where a kernel exchanges data between lanes through LDS and
                                                               relies on the SIMD's lockstep, its explicit wait is where the fibres re-align)
    asm volatile("" ...)                                    -> nothing
    asm volatile("ds_read_b64_tr_b16 %0, %1 [offset:..]")   -> hipemu transpose read            (a wave-level collective)
    __attribute__((address_space(N))), amdgpu_* attributes  -> dropped                          (LDS lives below 4 GB: a pointer IS its address)
    extern __shared__ T name[];                             -> T* name = workgroup's dynamic LDS
    __shared__ T name[..]..;                                -> reference into the workgroup's static LDS
"""
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.normpath(os.path.join(HERE, "..", ".."))
CSRC = os.path.normpath(os.path.join(HERE, "..", "..", "pytorch-studiogan_amd", "csrc"))


def _balanced(src, i):
    """index just past the parenthesis group that opens at src[i] == '('"""
    assert src[i] == "("
    depth, j, in_str = 0, i, False
    while j < len(src):
        c = src[j]
        if in_str:
            if c == "\\":
                j += 1
            elif c == '"':
                in_str = False
        elif c == '"':
            in_str = True
        elif c == "(":
            depth += 1
        elif c == ")":
            depth -= 1
            if depth == 0:
                return j + 1
        j += 1
    raise ValueError("unbalanced parenthesis")


def _split_top(s, sep):
    out, depth, cur, in_str = [], 0, "", False
    for c in s:
        if in_str:
            cur += c
            if c == '"':
                in_str = False
            continue
        if c == '"':
            in_str = True
        if c in "([":
            depth += 1
        elif c in ")]":
            depth -= 1
        if c == sep and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += c
    out.append(cur)
    return out


def _operands(section):
    """'"=&v"(lo), "v"(a)' -> ['lo', 'a']"""
    ops = []
    for item in _split_top(section, ","):
        item = item.strip()
        if not item:
            continue
        m = re.match(r'"[^"]*"\s*\((.*)\)\s*$', item, re.S)
        if not m:
            raise ValueError("asm operand: " + item)
        ops.append(m.group(1).strip())
    return ops


def _asm(stmt):
    """stmt = the text between 'asm volatile(' and its closing ')'"""
    parts = _split_top(stmt, ":")
    # '::' shows up as an empty part; normalise to template, outputs, inputs, clobbers
    tmpl = "".join(re.findall(r'"((?:[^"\\]|\\.)*)"', parts[0]))
    outs = _operands(parts[1]) if len(parts) > 1 else []
    ins = _operands(parts[2]) if len(parts) > 2 else []
    ops = outs + ins
    t = tmpl.strip()
    if t == "":
        return "((void)0)"
    m = re.match(r"s_waitcnt vmcnt\((%?)(\d+)\)$", t)
    if m:
        return "hipemu::waitcnt_vm(%s)" % (ops[int(m.group(2))] if m.group(1) else m.group(2))
    m = re.match(r"s_waitcnt lgkmcnt\((%?)(\d+)\)$", t)
    if m:
        return "hipemu::wave_lockstep(%s)" % (ops[int(m.group(2))] if m.group(1) else m.group(2))
    m = re.match(r"ds_read_b64_tr_b16 %0, %1(?: offset:(%?)(\d+))?$", t)
    if m:
        off = "0" if m.group(2) is None else (ops[int(m.group(2))] if m.group(1) else m.group(2))
        return "hipemu_tr_assign(%s, (%s) + (unsigned)(%s))" % (ops[0], ops[1], off)
    raise ValueError("untranslated asm: " + t)


_counter = [0]


def _shared(decl):
    """decl = text after '__shared__' up to (not including) ';'"""
    decl = re.sub(r"__attribute__\(\(aligned\(\d+\)\)\)", "", decl).strip()
    m = re.match(r"((?:unsigned\s+)?\w+)\s+(.*)$", decl, re.S)
    ty, rest = m.group(1), m.group(2)
    out = []
    for d in _split_top(rest, ","):
        d = d.strip()
        m = re.match(r"(\w+)\s*((?:\[[^\]]*\])*)$", d, re.S)
        name, dims = m.group(1), m.group(2)
        _counter[0] += 1
        k = _counter[0]
        out.append("typedef %s hipemu_sh_t%d%s; hipemu_sh_t%d& %s = *(hipemu_sh_t%d*)hipemu::static_lds(sizeof(hipemu_sh_t%d), %d)"
                   % (ty, k, dims, k, name, k, k, k))
    return "; ".join(out)


def translate(src, base=0):
    """base: first id of this file's static `__shared__` declarations (ids must not collide between the files of one kernel, and must not depend on
    what was translated before: the build cache is keyed on the translated text)"""
    _counter[0] = base
    src = re.sub(r"__attribute__\(\(address_space\(\d+\)\)\)", "", src)
    # __attribute__((amdgpu_...(...))) -- balanced: the arguments may hold parenthesised expressions
    out, i = "", 0
    for m in re.finditer(r"__attribute__\s*\(\(\s*amdgpu_\w+", src):
        if m.start() < i:
            continue
        j = _balanced(src, src.index("(", m.start()))
        out += src[i:m.start()]
        i = j
    src = out + src[i:]
    # extern __shared__ [attr] T name[];
    src = re.sub(r"extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s*)?(\w+)\s+(\w+)\s*\[\s*\]\s*;",
                 r"\1* \2 = (\1*)hipemu::dyn_lds();", src)
    # static __shared__ declarations
    out, i = "", 0
    for m in re.finditer(r"(?<![\w])__shared__\s", src):
        if m.start() < i:
            continue
        end = src.index(";", m.end())
        out += src[i:m.start()] + _shared(src[m.end():end])
        i = end
    src = out + src[i:]
    # inline assembly
    out, i = "", 0
    for m in re.finditer(r"\basm\s+volatile\s*\(", src):
        if m.start() < i:
            continue
        j = _balanced(src, m.end() - 1)
        out += src[i:m.start()] + _asm(src[m.end():j - 1])
        i = j
    src = out + src[i:]
    return src


PRELUDE = """// generated by tests/hipemu/translate.py -- do not edit
#include <hip/hip_runtime.h>
template <class T> static inline void hipemu_tr_assign(T& dst, unsigned addr) { hipemu::tr_read_deferred(dst, addr); }
"""


def source_files():
    """(name in the flat translated tree, path): csrc/*.{h,hip} and the units of csrc/ext/"""
    out = [(fn, os.path.join(CSRC, fn)) for fn in sorted(os.listdir(CSRC)) if fn.endswith(".h") or fn.endswith(".hip")]
    aug = os.path.join(CSRC, "ext")
    if os.path.isdir(aug):
        out += [(fn, os.path.join(aug, fn)) for fn in sorted(os.listdir(aug)) if fn.endswith(".h") or fn.endswith(".hip")]
    return out


def translate_tree(dst):
    os.makedirs(dst, exist_ok=True)
    files = source_files()
    assert len(set(n for n, _ in files)) == len(files), "file names must be unique across csrc/ and csrc/ext/"
    hdr = os.path.join(REPO, "include", "sgamd.h")
    for k, (fn, src_path) in enumerate(files):
        with open(src_path) as f:
            src = f.read()
        text = translate(src, 1000 * (1 + k))
        text = text.replace('"../../../include/sgamd.h"', '"%s"' % hdr).replace('"../../include/sgamd.h"', '"%s"' % hdr).replace('"../common.h"', '"common.h"')
        if fn == "common.h":
            text = text.replace("#pragma once", "#pragma once\n" + PRELUDE, 1)
        path = os.path.join(dst, fn)
        old = open(path).read() if os.path.exists(path) else None
        if old != text:
            with open(path, "w") as f:
                f.write(text)


if __name__ == "__main__":
    translate_tree(sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "_build", "src"))
