#!/bin/bash
# Collect HBM traffic of the dominant kernel for bench.py's roofline.traffic (run on the GPU box via gpurun):
#   separate rocprofv3 --pmc passes for FETCH_SIZE and WRITE_SIZE (they do not fit one pass, MI355X_MICROARCH §PMC),
#   kernel-trace only.  tools/traffic_json.py then applies the guide's gfx950 correction (FETCH_SIZE x2) and the KB unit.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
W=${1:-c2}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/traffic_${W}_$c -o t -- python3 $R/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/traffic_${W}_$c.log 2>&1
done
