"""C5 (issue, 2^20 credentials, the library's default secret mode) against the pass size (afx_ctx_set_chunk_items): the narrow tables of a
pass - two 96-byte entries per secret per-item term, re-read for every one of a chain's 128 additions - are 1 GB at 2^19 items and 126 MB at
2^16: does a pass whose tables fit the 256 MB Infinity Cache run the chains faster than one that streams them from HBM?"""
import ctypes as C, json, subprocess, sys, os
sys.path.insert(0, ".")
if len(sys.argv) > 1:
    import aeonflux_amd as afx
    chunk = int(sys.argv[1])
    orig = afx.Context.__init__
    def init(self, *a, **k):
        orig(self, *a, **k)
        if chunk:
            self.set_chunk_items(chunk)
    afx.Context.__init__ = init
    import bench
    sys.argv = ["bench.py", "--workload", sys.argv[2], "--steps", "5", "--warmup", "2", "--no-cpu-baseline"] + sys.argv[3:]
    bench.main()
else:
    for wl, extra in (("c5", []), ("show", []), ("c5", ["--secret-mode", "0"])):
        for chunk in (0, 1 << 18, 1 << 17, 1 << 16, 1 << 15):
            r = subprocess.run([sys.executable, __file__, str(chunk), wl] + extra, capture_output=True, text=True)
            try:
                d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
                k = d["roofline"]["kernels_ms_per_step"]
                print(wl, " ".join(extra), "chunk", chunk or "default (2^19)", round(d["value"]), "ms/step", round(d["ms_per_step"], 2), {x: round(v, 1) for x, v in k.items() if v > 2}, flush=True)
            except Exception as e:
                print("failed", wl, chunk, r.stderr[-300:])
