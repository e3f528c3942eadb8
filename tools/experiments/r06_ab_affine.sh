#!/bin/bash
# same box, alternating, on a build with tools/experiments/r06_affine_window_tables.patch applied: default (affine window tables from 2^17 items) vs AFX_VARIANT_CACHED_WINDOW_TABLES (0x80) through the python mirror's hook
for r in 1 2 3; do
  for v in 0 0x80; do
    AFX_TEST_PLAN_VARIANTS=$v python bench.py --workload ${1:-c3} --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --no-secondary --no-group-api --host-reps 1 ${2:-} 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; v = d['valu']
        print('variants $v', round(d['value']), 'ms/step', round(d['ms_per_step'], 3), 'MHz', round(v['core_clock_mhz_measured'], 1), 'per MHz', round(v['value_per_mhz'], 2), {k: x for k, x in r['kernels_ms_per_step'].items() if x > 1})"
  done
done
