#!/bin/bash
# quick kernel trace of one workload, grouped by kernel and grid.y:  tools/trace_quick.sh c5
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
w=${1:-c5}
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/tr_$w -o t -- python3 $R/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/tr_$w.log 2>&1
python3 $R/tools/rocpd_summary.py $(ls $R/gpurun_out/tr_$w/*/t_results.db $R/gpurun_out/tr_$w/t_results.db 2>/dev/null | head -1)
rm -rf $R/gpurun_out/tr_$w
