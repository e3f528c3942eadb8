#!/bin/bash
# launch-by-launch timelines of one 1-item issue / show / verify call (tools/small_call_timeline.py) into gpurun_out/timeline_<op>.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for op in "$@"; do
  rocprofv3 --kernel-trace --output-format csv -d $O/tl_$op -o tl -- python3 $R/tools/small_call_timeline.py $op > $O/tl_$op.log 2>&1
  f=$(find $O/tl_$op -name "*kernel_trace.csv" | head -1)
  (cd $R && python tools/small_call_timeline.py --parse $f > $O/timeline_$op.txt 2>&1)
  rm -rf $O/tl_$op
done
