#!/bin/bash
# HBM traffic of the multiscalar kernels with secret-independent addressing on (C5 issue, C3 verify): separate FETCH_SIZE /
# WRITE_SIZE passes with --kernel-trace only, as tools/collect_profiles.sh does for the default mode.
#   gpurun -- 'bash tools/collect_secret_mode_traffic.sh'   ->  gpurun_out/r03_secret_mode_traffic.txt, gpurun_out/r03_secret_traffic.json
# (copy both under profiles/: bench.py --secret-independent reads the json for roofline.traffic)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/secret_traffic
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
db() { ls $1/*/t_results.db $1/t_results.db 2>/dev/null | head -1; }
for w in c5 c3; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c -d $O/pmc_${w}_$c -o t -- python3 $R/bench.py --workload $w --secret-independent --steps 2 --warmup 1 --no-cpu-baseline --no-group-api > $O/pmc_${w}_$c.log 2>&1
  done
done
python3 $R/tools/traffic_json.py $R/gpurun_out/r03_secret_traffic.json \
  c5:$(db $O/pmc_c5_FETCH_SIZE):$(db $O/pmc_c5_WRITE_SIZE):$O/pmc_c5_FETCH_SIZE.log \
  c3:$(db $O/pmc_c3_FETCH_SIZE):$(db $O/pmc_c3_WRITE_SIZE):$O/pmc_c3_FETCH_SIZE.log > $R/gpurun_out/r03_secret_mode_traffic.txt 2>&1
for w in c5 c3; do grep '^{' $O/pmc_${w}_FETCH_SIZE.log | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$w bench line under the profiler: ms_per_step', round(d['ms_per_step'], 2), d['roofline']['kernels_ms_per_step'])" >> $R/gpurun_out/r03_secret_mode_traffic.txt; done
rm -rf $O/pmc_*_FETCH_SIZE $O/pmc_*_WRITE_SIZE
cat $R/gpurun_out/r03_secret_mode_traffic.txt
