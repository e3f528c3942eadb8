"""CPU-only: properties of the COMPILED gfx950 kernels that DESIGN.md states and that the compiler can silently take away.
Reads the device code object embedded in aeonflux_amd/lib/libaeonflux_gpu.so (llvm-objdump --offloading on a copy, llvm-readelf
--notes, llvm-objdump -d): nothing is compiled here, the test takes a few seconds.

 * register/scratch table of DESIGN.md section 3: the chain, table, decode and encode kernels are scratch-free, and the
   instances that are launched three blocks per CU fit 168 VGPRs;
 * secret-independent addressing: a secret scalar on a generator takes its table entry through a lane exchange (ds_bpermute_b32
   from lanes that each loaded one entry at an address made of the lane's id) - no load whose address comes from a digit, no LDS
   memory, no branch on the exec mask around it.  (Round 3 had found the compiler turning a select into a branch around a one-dword
   load, skipped when no lane of the wave held that digit - an access pattern that depended on the digits; hence this file.)"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "aeonflux_amd", "lib", "libaeonflux_gpu.so")
LLVM = "/opt/rocm/lib/llvm/bin"


@pytest.fixture(scope="module")
def code_object(tmp_path_factory):
    if not os.path.exists(LIB):
        pytest.fail("aeonflux_amd/lib/libaeonflux_gpu.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
    d = tmp_path_factory.mktemp("isa")
    lib = shutil.copy(LIB, d / "lib.so")
    subprocess.run([LLVM + "/llvm-objdump", "--offloading", str(lib)], check=True, capture_output=True, cwd=d)
    cos = [f for f in os.listdir(d) if "gfx950" in f]
    assert len(cos) == 1, os.listdir(d)
    co = str(d / cos[0])
    notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    kernels = {}
    for blk in notes.split(".args:")[1:] if ".args:" in notes else []:
        name = re.search(r"\.name:\s+(\S+)", blk)
        if name:
            kernels[name.group(1)] = {k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
                                      for k in ("vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size")}
    dis = subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
    bodies = {}
    for f in re.split(r"\n(?=[0-9a-f]{16} <)", dis):
        m = re.match(r"[0-9a-f]{16} <(\S+)>:", f)
        if m:
            bodies[m.group(1)] = f
    return kernels, bodies


def msm(kind, enc, sec):
    return "_Z5k_msmILi%dELb%dELb%dEEvPK12afx_msm_djobPKiS4_PiPjS6_jPy" % (kind, enc, sec)


def test_hot_kernels_are_scratch_free_and_fit_their_occupancy(code_object):
    kernels, _ = code_object
    assert len(kernels) >= 30, sorted(kernels)
    hot = [k for k in kernels if re.match(r"_Z\d+k_(msm|table_affine|decode|compress2x|negenc|pointsum|pointop|hash|from_uniform|scalarop)", k)]
    assert len(hot) >= 23, hot
    for k in hot:
        # k_from_uniform (two square-root chains inside Elligator) spills 31 dwords since its chains run in the 10-limb form; it
        # is faster with them than scratch-free on the 9-limb chains (3.16 -> 2.95 ms on C5, same box)
        assert kernels[k]["private_segment_fixed_size"] <= (128 if "k_from_uniform" in k else 0), (k, kernels[k])
    # launched with three blocks of 256 per CU (kernels.hip __launch_bounds__): 512 / 3 -> 168 registers
    three = [msm(0, 0, 0), msm(1, 0, 0), msm(2, 0, 0), msm(0, 0, 1)]
    for k in three:
        assert kernels[k]["vgpr_count"] <= 168, (k, kernels[k])
    for k in hot:
        assert kernels[k]["vgpr_count"] <= 256, (k, kernels[k])
    # the table kernels: one instance per table kind (one body with run-time layouts took 256 registers and scratch)
    tables = sorted(k for k in kernels if "k_msm_tables" in k)
    assert len(tables) == 4 and all(kernels[k]["vgpr_count"] <= 192 for k in tables), [(k, kernels[k]) for k in tables]


def test_secret_independent_lookups_address_nothing_by_a_digit(code_object):
    _, bodies = code_object
    count = lambda body, pat: len(re.findall(pat, body))
    for kind in (0, 1):
        for enc in (0, 1):
            sec, plain = bodies[msm(kind, enc, 1)], bodies[msm(kind, enc, 0)]
            # a secret scalar on a generator: the window's 32 multiples are loaded one per lane (seven 16-byte loads at an address made
            # of the lane's id) and the digit's multiple comes from the lane that holds it through ds_bpermute_b32, 27 dwords - a
            # lane exchange, no memory access; none of it in the ordinary instance
            assert count(plain, r"ds_bpermute_b32") == 0 and count(sec, r"ds_bpermute_b32") == 27, (kind, enc, count(sec, r"ds_bpermute_b32"))
            assert count(sec, r"global_load_dwordx[34]") >= count(plain, r"global_load_dwordx[34]") + 7
            # digit 0 takes the identity (entry 0 through scalar registers) with a mask insert per dword, not a branch
            assert count(sec, r"v_bfi_b32") >= 27
            # no LDS memory is read or written by these kernels: the exchange is the only DS instruction
            assert count(sec, r"\bds_(read|write|load|store)") == 0
            # no branch on the exec mask around the lookups: the SEC instance branches where its sibling does (+ the uniform
            # `secret` / `narrow` tests, which are scalar branches), not once per table word
            assert count(sec, r"s_cbranch_exec") <= count(plain, r"s_cbranch_exec") + 4, (kind, enc)
    # the narrow chain of the windowed SEC instance (a secret scalar on a per-item base): every stored entry of the lane's table is
    # read for every addition - two affine entries x six 16-byte loads, at the chain's two fetch sites
    assert count(bodies[msm(1, 0, 1)], r"global_load_dwordx4") >= 2 * 2 * 6


def test_the_four_wave_chain_kernel_exchanges_through_lds_only(code_object):
    """k_msm_quad (four waves per item chain, small passes): two LDS buffers of four field elements for 64 items, no scratch, a
    barrier per exchange that waits for the wave's LDS traffic only - the next table entry's global loads stay in flight across it
    (a `s_waitcnt vmcnt(0)` next to every barrier would put their latency back on the chain).  Its SEC instances (secret scalars):
    the lane exchange for generators - 18 limbs for the roles that multiply by (y+x)/2 or (y-x)/2, 9 for the one that takes dxy -
    and mask arithmetic where the digit and the sign choose among the loaded entries: no scratch, and no more branches on the
    exec mask than the uniform plan flags explain."""
    kernels, bodies = code_object
    quad = sorted(k for k in kernels if "k_msm_quad" in k)
    assert len(quad) == 4, quad
    psum = [k for k in kernels if "k_pointsum_quad" in k]
    assert len(psum) == 1 and kernels[psum[0]]["private_segment_fixed_size"] == 0 and kernels[psum[0]]["group_segment_fixed_size"] == 2 * 4 * 3 * 64 * 16, (psum, kernels.get(psum[0]) if psum else None)
    count = lambda body, pat: len(re.findall(pat, body))
    for k in quad:
        assert kernels[k]["private_segment_fixed_size"] == 0 and kernels[k]["group_segment_fixed_size"] == 2 * 4 * 3 * 64 * 16, (k, kernels[k])
        assert kernels[k]["vgpr_count"] <= 128, (k, kernels[k])
        body = bodies[k]
        barriers = count(body, r"s_barrier")
        assert barriers >= 6, (k, barriers)
        # at most the entry barrier (after the recoding) is a full __syncthreads; the exchanges wait on lgkmcnt alone
        full = count(body, r"s_waitcnt vmcnt\(0\)[^\n]*\n\s*s_barrier|s_waitcnt vmcnt\(0\) lgkmcnt\(0\)[^\n]*\n\s*s_barrier")
        assert full <= 2, (k, full, barriers)
        assert count(body, r"ds_read_b128|ds_load_b128") >= 12 and count(body, r"ds_write_b128|ds_store_b128") >= 3
        # the order of every exchange (kernels.hip quad_post: workgroup fences on the LDS address space around the barrier): the
        # role's store, a wait for this wave's LDS traffic, the barrier, and only then the loads of the other roles' elements - and
        # no LDS store between the barrier and those loads (the next exchange writes the other buffer, after them)
        ins = [ln.strip() for ln in body.split("\n")[1:] if ln.strip()]
        ordered = 0
        for i in [j for j, x in enumerate(ins) if x.startswith("s_barrier")]:
            w = max([j for j in range(i) if re.match(r"ds_(write|store)", ins[j])], default=-1)
            if w < 0:
                continue   # the entry barrier, before anything is exchanged
            waits = [j for j in range(w + 1, i) if re.match(r"s_waitcnt.*lgkmcnt\(0\)", ins[j])]
            r = min([j for j in range(i + 1, len(ins)) if re.match(r"ds_(read|load)", ins[j])], default=-1)
            w2 = min([j for j in range(i + 1, len(ins)) if re.match(r"ds_(write|store)", ins[j])], default=len(ins))
            assert waits and i < r < w2, (k, i, ins[max(0, i - 4):i + 4])
            ordered += 1
        assert ordered >= barriers - 1, (k, ordered, barriers)
    for rows in ("k_msm_quad_rows", "k_msm_quadI"):
        sec = next(bodies[k] for k in quad if rows in k and "ILb1E" in k)
        plain = next(bodies[k] for k in quad if rows in k and "ILb0E" in k)
        assert count(plain, r"ds_bpermute_b32") == 0 and count(sec, r"ds_bpermute_b32") == 27
        assert count(sec, r"s_cbranch_exec") <= count(plain, r"s_cbranch_exec") + 3, rows
