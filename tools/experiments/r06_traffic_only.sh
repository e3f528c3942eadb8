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
rm -rf $O/pmc_*_FETCH_SIZE $O/pmc_*_WRITE_SIZE
ls -la $O
