#!/bin/bash
# Everything profiles/ holds for one round, collected on the GPU box:  gpurun -- 'bash tools/collect_profiles.sh r02'
# Output: gpurun_out/profiles_<tag>/ (copy what should be judged into profiles/).
#  - bench JSON lines of the workloads (default = C3, C2, C5 issue, show)
#  - rocprofv3 --kernel-trace --stats summaries of the default bench command (C3) and of the C5 issue bench
#  - PMC passes, each in its own run with --kernel-trace only (MI355X_MICROARCH: separate passes):
#      FETCH_SIZE, WRITE_SIZE (HBM traffic per kernel, C3 and C5), the SQ issue/wait counters and the instruction mix (C3)
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/profiles_$TAG
mkdir -p $O
cd $R
python3 bench.py > $O/bench_c3.log 2>&1
grep '^{' $O/bench_c3.log | tail -1 > $O/${TAG}_bench_c3.json
for w in c2 c5 show; do
  python3 bench.py --workload $w > $O/bench_$w.log 2>&1
  grep '^{' $O/bench_$w.log | tail -1 > $O/${TAG}_bench_$w.json
done
cd /tmp && export TMPDIR=/tmp
db() { ls $1/*/$2_results.db $1/$2_results.db 2>/dev/null | head -1; }
for w in c3 c5; do
  rocprofv3 --kernel-trace --stats -d $O/trace_$w -o t -- python3 $R/bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline > $O/trace_$w.log 2>&1
  python3 $R/tools/rocpd_summary.py $(db $O/trace_$w t) > $O/${TAG}_${w}_kernel_trace.txt 2>&1
  grep '^{' $O/trace_$w.log | tail -1 > $O/${TAG}_${w}_kernel_trace_bench_line.json
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c -d $O/pmc_${w}_$c -o t -- python3 $R/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_${w}_$c.log 2>&1
  done
done
python3 $R/tools/traffic_json.py $O/${TAG}_traffic.json \
  c3:$(db $O/pmc_c3_FETCH_SIZE t):$(db $O/pmc_c3_WRITE_SIZE t):$O/pmc_c3_FETCH_SIZE.log \
  c5:$(db $O/pmc_c5_FETCH_SIZE t):$(db $O/pmc_c5_WRITE_SIZE t):$O/pmc_c5_FETCH_SIZE.log > $O/${TAG}_traffic.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d $O/pmc_sq -o t -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_sq.log 2>&1
python3 $R/tools/rocpd_pmc.py $(db $O/pmc_sq t) k_ > $O/${TAG}_c3_pmc.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O/pmc_mix -o t -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_mix.log 2>&1
python3 $R/tools/rocpd_pmc.py $(db $O/pmc_mix t) k_msm > $O/${TAG}_c3_instruction_mix.txt 2>&1
rm -rf $O/trace_c3 $O/trace_c5 $O/pmc_c3_FETCH_SIZE $O/pmc_c3_WRITE_SIZE $O/pmc_c5_FETCH_SIZE $O/pmc_c5_WRITE_SIZE $O/pmc_sq $O/pmc_mix   # databases are large; the summaries are kept
ls -la $O
