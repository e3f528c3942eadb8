import os, sys, subprocess, json
# each configuration in its own process (the knobs are read once)
for sl, first, ramp in ((0, 0, 0), (1 << 19, 1 << 16, 0), (1 << 19, 1 << 14, 4), (1 << 19, 1 << 15, 4), (1 << 19, 1 << 13, 4), (1 << 19, 1 << 15, 3), (1 << 19, 1 << 14, 2), (1 << 19, 1 << 16, 2), (1 << 18, 1 << 14, 4), (1 << 19, 1 << 16, 0)):
    env = dict(os.environ)
    if ramp: env["AFX_EXP_RAMP"] = str(ramp)
    if sl: env["AFX_EXP_SLICE"] = str(sl)
    if first: env["AFX_EXP_FIRST_SLICE"] = str(first)
    r = subprocess.run([sys.executable, "bench.py", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-secondary", "--no-group-api"], env=env, capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        c = d["config"]
        print("slice %7d first %6d ramp %d: value %.0f host %s (%.3f of value) wire %s" % (sl, first, ramp, d["value"], {k: round(v) for k, v in c["host_pointer_api_spread"].items()}, c["host_pointer_api_over_value"], {k: round(v) for k, v in c["wire_blob_api_spread"].items()}), flush=True)
    except Exception as e:
        print("failed", sl, first, r.stderr[-500:])
