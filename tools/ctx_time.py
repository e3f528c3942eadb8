import sys, time
sys.path.insert(0, '.')
import aeonflux_amd as afx, bench
for fx in ("c3_8attrs_SSPPeeee", "c5_16attrs"):
    params, key, ip = bench.load_fixture(fx)
    c = afx.Context(params, key, ip); c.close()
    t0 = time.perf_counter(); c = afx.Context(params, key, ip); dt = time.perf_counter() - t0; c.close()
    print(fx, "ctx create %.1f ms" % (dt * 1e3))
