#!/bin/bash
# Everything profiles/ holds for one round, collected on the GPU box:  gpurun -- 'tools/collect_profiles.sh r01'
# Output: gpurun_out/profiles_<tag>/ (copy what should be judged into profiles/).
#  - bench JSON lines of the four workloads
#  - rocprofv3 --kernel-trace --stats summary of the default bench command (C2)
#  - PMC passes, each in its own run with --kernel-trace only (MI355X_MICROARCH: separate passes):
#      FETCH_SIZE, WRITE_SIZE (HBM traffic of k_msm) and the SQ issue/wait counters
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/profiles_$TAG
mkdir -p $O
cd $R
for w in c2 c3 c5 show; do
  python3 bench.py --workload $w > $O/bench_$w.log 2>&1
  grep '^{' $O/bench_$w.log | tail -1 > $O/${TAG}_bench_$w.json
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/trace -o c2 -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/trace.log 2>&1
python3 $R/tools/rocpd_summary.py $(ls $O/trace/*/c2_results.db $O/trace/c2_results.db 2>/dev/null | head -1) > $O/${TAG}_c2_kernel_trace.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -o t -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_$c.log 2>&1
done
F=$(ls $O/pmc_FETCH_SIZE/*/t_results.db $O/pmc_FETCH_SIZE/t_results.db 2>/dev/null | head -1)
W=$(ls $O/pmc_WRITE_SIZE/*/t_results.db $O/pmc_WRITE_SIZE/t_results.db 2>/dev/null | head -1)
python3 $R/tools/traffic_json.py $O/${TAG}_traffic.json c2:$F:$W:3 > $O/traffic.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d $O/pmc_sq -o t -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_sq.log 2>&1
python3 $R/tools/rocpd_pmc.py $(ls $O/pmc_sq/*/t_results.db $O/pmc_sq/t_results.db 2>/dev/null | head -1) k_ > $O/${TAG}_c2_pmc.txt 2>&1
rm -rf $O/trace $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_sq   # databases are large; the summaries are what is kept
ls -la $O
