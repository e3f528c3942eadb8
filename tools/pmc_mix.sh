#!/bin/bash
# instruction-mix counters of the timed k_msm launches of the default bench (C2): tools/pmc_mix.sh  (on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $R/gpurun_out/pmc_mix -o t -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_mix.log 2>&1
python3 $R/tools/rocpd_pmc.py $(ls $R/gpurun_out/pmc_mix/*/t_results.db $R/gpurun_out/pmc_mix/t_results.db 2>/dev/null | head -1) --last 3 k_msm
rm -rf $R/gpurun_out/pmc_mix
