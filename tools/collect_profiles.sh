#!/bin/bash
# Everything profiles/ holds for one round, collected on the GPU box:  gpurun -- 'bash tools/collect_profiles.sh r06'
# Output: gpurun_out/profiles_<tag>/ (copy what should be judged into profiles/).
#  - bench JSON lines: default = C3; C2; C5 issue and show in the library's default mode (secrets off the table addresses on the
#    prover-side calls, afx_ctx_set_secret_independent_addressing 2) and with the fast tables (mode 0, rounds 1-3); C3 in mode 1
#  - rocprofv3 --kernel-trace --stats summaries of the default bench command (C3) and of the C5 issue bench (both modes)
#  - PMC passes, each in its own run with --kernel-trace only (MI355X_MICROARCH: separate passes):
#      FETCH_SIZE, WRITE_SIZE -> <tag>_traffic.json (C3 default; C5 mode 0) and <tag>_secret_traffic.json (C5 default mode; C3
#      mode 1) with the sha256 of the kernel sources they were measured on; the SQ issue/wait counters and the instruction mix (C3),
#      the SQ counters of C5 in both modes
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/profiles_$TAG
mkdir -p $O
cd $R
line() { grep '^{' $1 | tail -1; }
# the profiled runs measure the workload's own timed steps: no secondary workloads, no group leg, one repetition of the host legs
Q="--no-cpu-baseline --no-secondary --no-group-api --host-reps 1"
cd /tmp && export TMPDIR=/tmp
db() { ls $1/*/t_results.db $1/t_results.db 2>/dev/null | head -1; }
trace() {   # name, bench flags...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats -d $O/trace_$name -o t -- python3 $R/bench.py "$@" --steps 10 --warmup 2 $Q > $O/trace_$name.log 2>&1
  python3 $R/tools/rocpd_summary.py $(db $O/trace_$name) > $O/${TAG}_${name}_kernel_trace.txt 2>&1
  line $O/trace_$name.log > $O/${TAG}_${name}_kernel_trace_bench_line.json
  rm -rf $O/trace_$name
}
pmc2() {    # name, bench flags...: FETCH_SIZE and WRITE_SIZE passes
  local name=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c -d $O/pmc_${name}_$c -o t -- python3 $R/bench.py "$@" --steps 3 --warmup 1 $Q > $O/pmc_${name}_$c.log 2>&1
  done
}
trace c3 --workload c3
trace c5 --workload c5
trace c5_fast_tables --workload c5 --secret-mode 0
pmc2 c3 --workload c3
pmc2 c5fast --workload c5 --secret-mode 0
pmc2 c5 --workload c5
pmc2 c3all --workload c3 --secret-mode 1
pmc2 c2 --workload c2
pmc2 show --workload show
pmc2 showfast --workload show --secret-mode 0
python3 $R/tools/traffic_json.py $O/${TAG}_traffic.json \
  c3:$(db $O/pmc_c3_FETCH_SIZE):$(db $O/pmc_c3_WRITE_SIZE):$O/pmc_c3_FETCH_SIZE.log \
  c5:$(db $O/pmc_c5fast_FETCH_SIZE):$(db $O/pmc_c5fast_WRITE_SIZE):$O/pmc_c5fast_FETCH_SIZE.log \
  c2:$(db $O/pmc_c2_FETCH_SIZE):$(db $O/pmc_c2_WRITE_SIZE):$O/pmc_c2_FETCH_SIZE.log \
  show:$(db $O/pmc_showfast_FETCH_SIZE):$(db $O/pmc_showfast_WRITE_SIZE):$O/pmc_showfast_FETCH_SIZE.log > $O/${TAG}_traffic.txt 2>&1
python3 $R/tools/traffic_json.py $O/${TAG}_secret_traffic.json \
  c5:$(db $O/pmc_c5_FETCH_SIZE):$(db $O/pmc_c5_WRITE_SIZE):$O/pmc_c5_FETCH_SIZE.log \
  c3:$(db $O/pmc_c3all_FETCH_SIZE):$(db $O/pmc_c3all_WRITE_SIZE):$O/pmc_c3all_FETCH_SIZE.log \
  show:$(db $O/pmc_show_FETCH_SIZE):$(db $O/pmc_show_WRITE_SIZE):$O/pmc_show_FETCH_SIZE.log > $O/${TAG}_secret_traffic.txt 2>&1
sq() {      # name, bench flags...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES -d $O/pmc_sq_$name -o t -- python3 $R/bench.py "$@" --steps 3 --warmup 1 $Q > $O/pmc_sq_$name.log 2>&1
  python3 $R/tools/rocpd_pmc.py $(db $O/pmc_sq_$name) k_ > $O/${TAG}_${name}_pmc.txt 2>&1
  rm -rf $O/pmc_sq_$name
}
sq c3 --workload c3
sq c5 --workload c5
sq c5_fast_tables --workload c5 --secret-mode 0
# the lane exchange of the secret generator lookups: LDS bank conflicts of the C5 issue bench (default mode), and the microbenchmark
# of ds_bpermute_b32 against the pattern of lanes read, plain and under the counters (one launch per pattern, in pattern order)
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -d $O/pmc_lds_c5 -o t -- python3 $R/bench.py --workload c5 --steps 3 --warmup 1 $Q > $O/pmc_lds_c5.log 2>&1
python3 $R/tools/rocpd_pmc.py $(db $O/pmc_lds_c5) k_msm > $O/${TAG}_c5_lds_pmc.txt 2>&1
rm -rf $O/pmc_lds_c5
if [ ! -x $R/variants/bperm_lookup ]; then mkdir -p $R/variants; hipcc -O3 -Wno-unused-value --offload-arch=gfx950 $R/tools/ubench/bperm_lookup.hip -o $R/variants/bperm_lookup > /dev/null 2>&1; fi
$R/variants/bperm_lookup > $O/${TAG}_bperm_lookup.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/pmc_bperm -o t -- $R/variants/bperm_lookup once > /dev/null 2>&1
python3 $R/tools/rocpd_pmc.py $(db $O/pmc_bperm) --each k_lookup >> $O/${TAG}_bperm_lookup.txt 2>&1
rm -rf $O/pmc_bperm
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O/pmc_mix -o t -- python3 $R/bench.py --steps 3 --warmup 1 $Q > $O/pmc_mix.log 2>&1
python3 $R/tools/rocpd_pmc.py $(db $O/pmc_mix) k_msm > $O/${TAG}_c3_instruction_mix.txt 2>&1
rm -rf $O/pmc_*_FETCH_SIZE $O/pmc_*_WRITE_SIZE $O/pmc_mix   # databases are large; the summaries are kept
# the bench lines last: their roofline.traffic comes from the traffic files just measured on these kernels (bench.py takes the newest
# profiles/rNN_traffic.json whose kernel_sources_sha256 is this tree's)
mkdir -p $R/profiles && cp $O/${TAG}_traffic.json $O/${TAG}_secret_traffic.json $R/profiles/
cd $R
python3 bench.py > $O/bench_c3.log 2>&1; line $O/bench_c3.log > $O/${TAG}_bench_c3.json
python3 bench.py --workload c2 > $O/bench_c2.log 2>&1; line $O/bench_c2.log > $O/${TAG}_bench_c2.json
for w in c5 show; do
  python3 bench.py --workload $w > $O/bench_$w.log 2>&1; line $O/bench_$w.log > $O/${TAG}_bench_$w.json
  python3 bench.py --workload $w --secret-mode 0 --no-cpu-baseline > $O/bench_${w}_mode0.log 2>&1; line $O/bench_${w}_mode0.log > $O/${TAG}_bench_${w}_fast_tables.json
done
python3 bench.py --secret-mode 1 --no-cpu-baseline > $O/bench_c3_mode1.log 2>&1; line $O/bench_c3_mode1.log > $O/${TAG}_bench_c3_secret_everywhere.json
# the small-call and mixed-request measurements of the round
cd $R
python3 tools/mixed_concurrency.py 64 16 > $O/${TAG}_mixed_concurrency.txt 2>&1
for a in "8 16" "64 1" "32 64" "64 256"; do python3 tools/mixed_concurrency.py $a >> $O/${TAG}_mixed_concurrency.txt 2>&1; done
python3 tools/small_call_latency.py > $O/${TAG}_small_call_latency.txt 2>&1
python3 tools/midsize_host_calls.py > $O/${TAG}_midsize_host_calls.txt 2>&1
# concurrent small calls on ONE context (round 5): K threads x 1-item calls through the native driver
python3 tools/concurrent_small_calls.py --one-context --threads 1,2,4,8,16,32,64,128,256 > $O/${TAG}_coalesced_calls.txt 2>&1
# one 1-item call of each prover / verifier operation, launch by launch
tools/timeline.sh issue show verify
(echo "# tools/timeline.sh issue show verify: rocprofv3 --kernel-trace of ONE 1-item host-pointer call each (issue: 16 attributes; show, verify: the C3 shape), launch by launch: start offset, duration, gap to the previous launch"; cat $R/gpurun_out/timeline_issue.txt $R/gpurun_out/timeline_show.txt $R/gpurun_out/timeline_verify.txt) > $O/${TAG}_small_call_timeline.txt
# the launch path of N ranks on this one device (not a scaling figure): RCCL refuses two ranks on one GPU, so the ranks agree on gloo
AFX_BENCH_DEVICE=0 python3 bench.py --gpus 8 --steps 3 --warmup 1 --batch 32768 --no-cpu-baseline --no-group-api 2> $O/bench_gpus8.err | grep '^{' | tail -1 > $O/${TAG}_bench_gpus8_one_device_gloo.json
python3 bench.py --preflight --gpus 1 > $O/${TAG}_preflight.json 2>&1
ls -la $O
