#!/bin/bash
# Compare builds of the library on the same GPU box: tools/ab_bench.sh <workload> <rounds> <variant.so>...
# BENCH_FLAGS: extra bench.py flags (e.g. --secret-independent).  Alternates the default build and each variant so that box-to-box and thermal differences cancel.
set -u
WL=$1; ROUNDS=$2; shift 2
LIB=aeonflux_amd/lib/libaeonflux_gpu.so
cp $LIB /tmp/ab_default.so
for r in $(seq $ROUNDS); do
  for which in /tmp/ab_default.so "$@"; do
    cp $which $LIB
    python bench.py --workload $WL --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline ${BENCH_FLAGS:-} 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        v = d.get('valu') or {}
        print('%-40s' % '$which', round(d['value']), 'ms/step', round(d['ms_per_step'], 3), 'MHz', round(v.get('core_clock_mhz_measured') or 0, 1), 'per MHz', round(v.get('value_per_mhz') or 0, 2),
              'k_msm*', round(r['kernel_ms_per_step'], 3), r.get('kernels_ms_per_step', r.get('other_kernels_ms_per_step')))"
  done
done
cp /tmp/ab_default.so $LIB
