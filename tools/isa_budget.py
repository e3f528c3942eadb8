#!/usr/bin/env python3
"""The instruction budget of the field and point operations, by class, from the compiled gfx950 ISA (no GPU needed).

One kernel per operation, compiled from the SHIPPING headers (fe.cuh, ge.cuh) with the shipping flags: operands come from memory,
results go to memory, and a kernel that only loads and stores is subtracted.  The classes are those of the issue price list
(profiles/r01_valu_rates_ubench.txt): multiply-adds (v_mad_i64_i32 / v_mad_u64_u32, ~5.4 cycles per wave-instruction), other 64-bit
and full-rate-half VALU (v_ashrrev_i64, v_lshl_add_u64, v_alignbit, v_mul_*: ~4.2), simple 32-bit VALU (and / add / sub / shift /
mov / cndmask: ~2.4).  Also counts DEAD multiply-adds: a v_mad with a zero addend whose result is overwritten unread - what the
pinned first product of a chain compiled to until round 6 (fe.cuh fe_mul_impl).

    python tools/isa_budget.py > profiles/r06_isa_budget.txt
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "aeonflux_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"

SRC = r'''
#include "ge.cuh"
struct io { int32_t v[9 * 16]; };
#define LD(k) ld(in, k, t)
__device__ __forceinline__ fe ld(const int32_t* in, int k, uint32_t t) { fe r; for (int i = 0; i < 9; i++) r.v[i] = in[(k * 9 + i) * 65536 + t]; return r; }
__device__ __forceinline__ void st(int32_t* out, int k, uint32_t t, const fe& f) { for (int i = 0; i < 9; i++) out[(k * 9 + i) * 65536 + t] = f.v[i]; }
#define KERNEL(name, nin, ...) extern "C" __global__ void __launch_bounds__(256) name(const int32_t* in, int32_t* out, int neg) { const uint32_t t = blockIdx.x * 256 + threadIdx.x; __VA_ARGS__ }
// baselines: load n field elements, store m
KERNEL(base_2_1, 2, { fe a = LD(0), b = LD(1); st(out, 0, t, fe_add(a, b)); })
KERNEL(op_fe_mul, 2, { fe a = LD(0), b = LD(1); st(out, 0, t, fe_mul(a, b)); })
KERNEL(op_fe_mul_raw, 2, { fe a = LD(0), b = LD(1); st(out, 0, t, fe_mul_raw(a, b)); })
KERNEL(base_1_1, 1, { fe a = LD(0); st(out, 0, t, a); })
KERNEL(op_fe_sq, 1, { fe a = LD(0); st(out, 0, t, fe_sq(a)); })
KERNEL(op_fe_sq_raw, 1, { fe a = LD(0); st(out, 0, t, fe_sq_raw(a)); })
KERNEL(base_3_4, 3, { fe a = LD(0), b = LD(1), c = LD(2); st(out, 0, t, a); st(out, 1, t, b); st(out, 2, t, c); st(out, 3, t, fe_add(a, b)); })
KERNEL(op_ge_p2_dbl, 3, { ge_p2 p; p.X = LD(0); p.Y = LD(1); p.Z = LD(2); ge_p1p1 r = ge_p2_dbl(p); st(out, 0, t, r.X); st(out, 1, t, r.Y); st(out, 2, t, r.Z); st(out, 3, t, r.T); })
KERNEL(op_dbl_and_to_p2, 3, { ge_p2 p; p.X = LD(0); p.Y = LD(1); p.Z = LD(2); ge_p2 r = ge_p1p1_to_p2_before_dbl(ge_p2_dbl(p)); st(out, 0, t, r.X); st(out, 1, t, r.Y); st(out, 2, t, r.Z); st(out, 3, t, r.X); })
KERNEL(base_8_4, 8, { fe a = LD(0), b = LD(1), c = LD(2), d = LD(3), e = LD(4), f = LD(5), g = LD(6), h = LD(7); st(out, 0, t, fe_add(a, e)); st(out, 1, t, fe_add(b, f)); st(out, 2, t, fe_add(c, g)); st(out, 3, t, fe_add(d, h)); })
KERNEL(op_ge_add_cached, 8, { ge_p3 p; p.X = LD(0); p.Y = LD(1); p.Z = LD(2); p.T = LD(3); ge_cached q; q.YpX = LD(4); q.YmX = LD(5); q.Z2 = LD(6); q.T2d = LD(7);
  ge_p1p1 r = ge_add_cached(p, q, neg != 0); st(out, 0, t, r.X); st(out, 1, t, r.Y); st(out, 2, t, r.Z); st(out, 3, t, r.T); })
KERNEL(op_add_cached_and_to_p3, 8, { ge_p3 p; p.X = LD(0); p.Y = LD(1); p.Z = LD(2); p.T = LD(3); ge_cached q; q.YpX = LD(4); q.YmX = LD(5); q.Z2 = LD(6); q.T2d = LD(7);
  ge_p3 r = ge_p1p1_to_p3_for<GE_FOR_ADD>(ge_add_cached(p, q, neg != 0)); st(out, 0, t, r.X); st(out, 1, t, r.Y); st(out, 2, t, r.Z); st(out, 3, t, r.T); })
KERNEL(base_7_4, 7, { fe a = LD(0), b = LD(1), c = LD(2), d = LD(3), e = LD(4), f = LD(5), g = LD(6); st(out, 0, t, fe_add(a, e)); st(out, 1, t, fe_add(b, f)); st(out, 2, t, fe_add(c, g)); st(out, 3, t, d); })
KERNEL(op_ge_madd, 7, { ge_p3 p; p.X = LD(0); p.Y = LD(1); p.Z = LD(2); p.T = LD(3); ge_niels q; q.ypx = LD(4); q.ymx = LD(5); q.xyd = LD(6);
  ge_p1p1 r = ge_madd(p, q, neg != 0); st(out, 0, t, r.X); st(out, 1, t, r.Y); st(out, 2, t, r.Z); st(out, 3, t, r.T); })
'''

MAD = ("v_mad_i64_i32", "v_mad_u64_u32")
WIDE = ("v_ashrrev_i64", "v_lshrrev_b64", "v_lshlrev_b64", "v_lshl_add_u64", "v_alignbit_b32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32", "v_mad_i32_i24",
        "v_mul_i32_i24", "v_mul_hi_i32_i24", "v_mad_u32_u24", "v_add3_u32", "v_lshl_add_u32", "v_and_or_b32", "v_bfe_i32", "v_bfe_u32", "v_mul_u32_u24", "v_lshl_or_b32", "v_add_lshl_u32")


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def functions(dis):
    out, cur = {}, None
    for l in dis.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(\w+)>:", l)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        l = re.sub(r"\s*//.*", "", l).strip()
        if cur is not None and l and not l.startswith("s_nop") and not l.startswith("s_code_end"):
            cur.append(l)
    return out


def classify(body):
    c = collections.Counter()
    for i, l in enumerate(body):
        op = l.split()[0]
        base = re.sub(r"_e(32|64)$", "", op)
        if base in MAD:
            c["mad"] += 1
            parts = [p.strip() for p in l.split(None, 1)[1].split(",")]
            if parts[-1] == "0":
                dst, live = regs(parts[0]), None
                for l2 in body[i + 1:i + 400]:
                    if l2.startswith("s_cbranch") or l2.startswith("s_branch") or l2.startswith("s_endpgm"):
                        live = True
                        break
                    if " " not in l2:
                        continue
                    ps = [p.strip() for p in l2.split(None, 1)[1].split(",")]
                    srcs = set()
                    for p in ps[1:]:
                        srcs |= regs(p)
                    if l2.startswith(("flat_store", "global_store", "buffer_store", "ds_")):
                        srcs |= regs(ps[0])
                    if srcs & dst:
                        live = True
                        break
                    if regs(ps[0]) >= dst:
                        live = False
                        break
                if live is False:
                    c["dead_mad"] += 1
        elif base in WIDE:
            c["wide"] += 1
            c["op:" + base] += 1
        elif op.startswith("v_"):
            c["simple"] += 1
            c["op:" + base] += 1
        elif op.startswith(("global_", "flat_", "buffer_", "ds_", "scratch_")):
            c["mem"] += 1
        else:
            c["scalar"] += 1
    return c


def main():
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "budget.hip")
        open(src, "w").write(SRC)
        co = os.path.join(d, "budget.co")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "--genco", "-I", CSRC, src, "-o", co], check=True)
        elf = os.path.join(d, "budget.elf")   # (--genco writes an offload bundle: the gfx950 ELF is unbundled from it)
        subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + co, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + elf], check=True)
        dis = subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", elf], check=True, capture_output=True, text=True).stdout
    fn = {k: classify(v) for k, v in functions(dis).items()}
    print("# tools/isa_budget.py: instructions per lane and operation in the compiled gfx950 code (shipping fe.cuh / ge.cuh, hipcc -O3), by issue class,")
    print("# a load/store-only kernel of the same shape subtracted.  mad: v_mad_i64_i32 + v_mad_u64_u32 (~5.4 cycles per wave-instruction);")
    print("# wide: 64-bit shifts / adds, v_alignbit, 24-bit multiplies (~4.2); simple: and / add / sub / 32-bit shifts / mov / cndmask (~2.4); cycles = 5.4 mad + 4.2 wide + 2.4 simple")
    print("%-26s %6s %6s %6s %6s %8s %9s   %s" % ("operation", "mad", "dead", "wide", "simple", "cycles", "mad share", "the non-mad instructions"))
    pairs = [("op_fe_mul", "base_2_1"), ("op_fe_mul_raw", "base_2_1"), ("op_fe_sq", "base_1_1"), ("op_fe_sq_raw", "base_1_1"), ("op_ge_p2_dbl", "base_3_4"),
             ("op_dbl_and_to_p2", "base_3_4"), ("op_ge_add_cached", "base_8_4"), ("op_add_cached_and_to_p3", "base_8_4"), ("op_ge_madd", "base_7_4")]
    for op, base in pairs:
        a, b = fn[op], fn[base]
        mad, dead = a["mad"] - b["mad"], a["dead_mad"]
        wide, simple = a["wide"] - b["wide"], a["simple"] - b["simple"]
        cyc = 5.4 * mad + 4.2 * wide + 2.4 * simple
        detail = collections.Counter({k[3:]: v - b.get(k, 0) for k, v in a.items() if k.startswith("op:")})
        detail = ", ".join("%s %d" % (k, v) for k, v in detail.most_common() if v > 0)
        print("%-26s %6d %6d %6d %6d %8.0f %8.1f%%   %s" % (op[3:], mad, dead, wide, simple, cyc, 100 * 5.4 * mad / cyc, detail))


if __name__ == "__main__":
    main()
