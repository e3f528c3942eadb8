"""CPU-only checks of the product's host side: the C-ABI library loads and exports every symbol that
include/aeonflux_gpu.h declares, refuses to run without a GPU (no CPU fallback), and the multi-GPU sharding
logic partitions/gathers correctly over a 2-process gloo group."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "aeonflux_gpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(afx_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import aeonflux_amd as afx
    names = declared_functions()
    assert len(names) >= 20 and "afx_verify_presentations_dev" in names and "afx_issue" in names and "afx_show" in names
    lib = afx.lib()
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_header_is_valid_c_and_the_c_example_links(tmp_path):
    """the boundary is a C ABI: the header must compile as C99 (not only as C++), and the C example must link against the library"""
    hdr = os.path.join(ROOT, "include")
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-I", hdr, "-x", "c", os.path.join(hdr, "aeonflux_gpu.h")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    import aeonflux_amd as afx
    main = tmp_path / "main.c"
    main.write_text(r'''#include "aeonflux_gpu.h"
int verify_my_range(afx_ctx*, unsigned, unsigned, const afx_shape*, const afx_presentation_soa*, size_t, unsigned char*);
int verify_request_stream(afx_ctx*, const unsigned char*, size_t, unsigned char*, size_t, size_t*);
typedef int (*afx_cpu_verify_fn)(void*, const unsigned char*, size_t, unsigned char*, size_t, size_t*);
int verify_request_stream_or_fall_through(afx_ctx*, afx_cpu_verify_fn, void*, const unsigned char*, size_t, unsigned char*, size_t, size_t*, unsigned long*);
/* a stand-in CPU verifier: answers 3 presentations, all accepted */
static int stub_cpu_verify(void* issuer, const unsigned char* s, size_t l, unsigned char* st, size_t cap, size_t* n) {
  (void)issuer; (void)s; (void)l;
  if (cap < 3) return AFX_E_BAD_ARGS;
  st[0] = st[1] = st[2] = AFX_ST_OK; *n = 3; return AFX_OK;
}
int main(void) {
  size_t n = 7;
  unsigned long fell = 0;
  unsigned char st[4] = { 9, 9, 9, 9 };
  if (verify_my_range(0, 1, 0, 0, 0, 0, 0) != AFX_E_BAD_ARGS || verify_request_stream(0, 0, 0, 0, 0, &n) != AFX_E_BAD_ARGS) return 1;
  /* no engine (a box whose GPU is gone): the request is answered by the CPU verifier, and counted */
  if (verify_request_stream_or_fall_through(0, stub_cpu_verify, 0, (const unsigned char*)"x", 1, st, 4, &n, &fell) != AFX_OK) return 2;
  if (n != 3 || fell != 1 || st[0] != AFX_ST_OK || st[2] != AFX_ST_OK || st[3] != 9) return 3;
  /* ... and without a CPU verifier the fault itself comes back: never a verdict */
  if (verify_request_stream_or_fall_through(0, 0, 0, (const unsigned char*)"x", 1, st, 4, &n, &fell) != AFX_E_NO_DEVICE || fell != 2) return 4;
  return 0;
}
''')
    exe = tmp_path / "example"
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", hdr, os.path.join(ROOT, "integration", "example_verify.c"), str(main),
                        "-L", os.path.dirname(afx.LIB_PATH), "-laeonflux_gpu", "-Wl,-rpath," + os.path.dirname(afx.LIB_PATH), "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # null arguments are refused before any device is touched, so this runs without a GPU
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout, r.stderr)


def test_no_cpu_fallback_without_gpu():
    """on a box without a HIP device the engine must fail loudly, never compute on the CPU"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import aeonflux_amd as afx
    import json
    r = json.load(open(os.path.join(ROOT, "tests", "golden", "flows.json")))["flows"][0]
    with pytest.raises(afx.AfxError) as e:
        afx.Context(bytes.fromhex(r["params"]), bytes.fromhex(r["key"]), bytes.fromhex(r["issuer_params"]))
    assert e.value.rc == afx.E_NO_DEVICE


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "aeonflux_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip", ".cuh")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "afx_oracle" not in text and "afxo_" not in text, f


def test_shard_bounds_cover_everything():
    from aeonflux_amd.sharding import shard_bounds
    for count in (0, 1, 7, 8, 65536, 65537, (1 << 22) + 3):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(count, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == count
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


WORKER = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np
import torch.distributed as dist
from aeonflux_amd.sharding import verify_sharded, shard_bounds
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
count = 1001
rng = np.random.default_rng(5)
pres = {"challenge": rng.integers(0, 256, (count, 32), dtype=np.uint8), "C_y": rng.integers(0, 256, (3, count, 32), dtype=np.uint8),
        "enc": [{"E1": rng.integers(0, 256, (count, 32), dtype=np.uint8), "responses": rng.integers(0, 256, (6, count, 32), dtype=np.uint8)}]}
calls = []
def fake_verify(shape, shard):   # stands in for the GPU engine: status is a function of the item's own bytes
    calls.append(shard["challenge"].shape[0])
    assert shard["C_y"].shape == (3, calls[-1], 32) and shard["enc"][0]["responses"].shape == (6, calls[-1], 32)
    return (shard["challenge"][:, 0] ^ shard["C_y"][2, :, 5] ^ shard["enc"][0]["E1"][:, 9]) & 1
full = verify_sharded(fake_verify, None, pres, count, rank, world)
want = (pres["challenge"][:, 0] ^ pres["C_y"][2, :, 5] ^ pres["enc"][0]["E1"][:, 9]) & 1
lo, hi = shard_bounds(count, world, rank)
assert calls == [hi - lo] and np.array_equal(full, want), (calls, rank)
dist.barrier()
dist.destroy_process_group()
print("ok", rank)
"""


def test_sharded_verify_two_processes_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "ok 0" in outs[0] and "ok 1" in outs[1]
