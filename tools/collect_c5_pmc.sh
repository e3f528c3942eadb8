#!/bin/bash
# SQ issue/wait counters per kernel for the C5 issue bench (tools/collect_profiles.sh collects them for C3).
#   gpurun -- 'bash tools/collect_c5_pmc.sh [extra bench flags]'  ->  gpurun_out/c5_pmc/c5_pmc.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/c5_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
db() { ls $1/*/t_results.db $1/t_results.db 2>/dev/null | head -1; }
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES -d $O/pmc_sq -o t -- python3 $R/bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline "$@" > $O/pmc_sq.log 2>&1
python3 $R/tools/rocpd_pmc.py $(db $O/pmc_sq) k_ > $O/c5_pmc.txt 2>&1
rm -rf $O/pmc_sq
grep -A8 "k_msm<" $O/c5_pmc.txt | head -120
