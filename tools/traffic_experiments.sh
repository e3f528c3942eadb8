#!/bin/bash
# Table-traffic experiments: for the shipped build and each experiment build in variants/ (built by tools/build_variant.sh from
# tools/experiments/r02_table_traffic_switches.patch with
# -DAFX_EXPERIMENT_*; the switches are not in the shipping sources), the per-kernel time of C3 verification and C5 issuance and the HBM traffic of the MSM
# kernels (separate FETCH_SIZE / WRITE_SIZE passes).  Run on the GPU box:  bash tools/traffic_experiments.sh <variant.so>...
# Output: gpurun_out/traffic_experiments.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/texp
mkdir -p $O
LIB=$R/aeonflux_amd/lib/libaeonflux_gpu.so
cp $LIB /tmp/shipped.so
export AFX_BENCH_UNCHECKED=1
cd /tmp && export TMPDIR=/tmp
db() { ls $1/*/t_results.db $1/t_results.db 2>/dev/null | head -1; }
OUT=$R/gpurun_out/traffic_experiments.txt
: > $OUT
for which in /tmp/shipped.so "$@"; do
  [ "$which" = /tmp/shipped.so ] || which=$R/$which
  cp $which $LIB
  name=$(basename $which .so)
  for w in c3 c5; do
    python3 $R/bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > $O/${name}_$w.log 2>&1
    for c in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --kernel-trace --pmc $c -d $O/${name}_${w}_$c -o t -- python3 $R/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline > $O/${name}_${w}_$c.log 2>&1
    done
    echo "== $name $w" >> $OUT
    python3 - >> $OUT <<PY
import json
d = json.loads([l for l in open("$O/${name}_$w.log") if l.startswith("{")][-1])
print("value %.0f /s  ms_per_step %.2f  kernels_ms_per_step %s" % (d["value"], d["ms_per_step"], {k: v for k, v in d["roofline"]["kernels_ms_per_step"].items() if k.startswith("k_msm")}))
PY
    python3 $R/tools/traffic_json.py $O/${name}_$w.json $w:$(db $O/${name}_${w}_FETCH_SIZE):$(db $O/${name}_${w}_WRITE_SIZE):$O/${name}_${w}_FETCH_SIZE.log 2>&1 | grep k_msm >> $OUT
    rm -rf $O/${name}_${w}_FETCH_SIZE $O/${name}_${w}_WRITE_SIZE
  done
done
cp /tmp/shipped.so $LIB
cat $OUT
