#!/bin/bash
# Compare two whole TREES of this repository on the same GPU box (library + its own bench.py, so that ABI changes between
# rounds do not matter): tools/ab_trees.sh <workload> <rounds> <other tree> [label]
# The other tree is a checkout of an earlier commit with its library built, e.g.
#   mkdir -p variants/r02_tree && git archive 5650140 | tar -x -C variants/r02_tree && make -C variants/r02_tree/aeonflux_amd/csrc
# (variants/ is git-ignored and travels to the GPU box).  Alternates the two so that box-to-box and thermal differences cancel.
set -u
WL=$1; ROUNDS=$2; OTHER=$3; LABEL=${4:-other}
HERE=$(pwd)
line() {
  python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('%-10s' % '$1', round(d['value']), 'items/s  ms/step', round(d['ms_per_step'], 3), ' clock MHz', d.get('valu', {}).get('core_clock_mhz_measured'), ' kernels', r.get('kernels_ms_per_step', r.get('other_kernels_ms_per_step')))"
}
for r in $(seq $ROUNDS); do
  (cd $HERE && python bench.py --workload $WL --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline 2>/dev/null | line HEAD)
  (cd $OTHER && python bench.py --workload $WL --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline 2>/dev/null | line $LABEL)
done
