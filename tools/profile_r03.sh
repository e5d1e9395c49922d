#!/bin/bash
# Round-3 profile collection (run on the MI355X box through gpurun, from the repository root):
#   rocprofv3 --kernel-trace --stats of the bench workloads and of the per-K sweeps, then the two
#   --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, no tracing beside them).  Summaries land in
#   gpurun_out/prof_r03/; tools/collect_profiles.py copies what is judged into profiles/.
# rocprofv3-safe: the profiled PROGRAM comes directly after `--` (python3 <script> ...) -- never `env`, `bash -c`, a
# `#!/usr/bin/env` script or a re-exec'ing launcher: the profiler's preloaded library initialises the GPU before the
# program starts, and an exec from a GPU-initialised process takes the box down (the pool refuses it).
set -o pipefail
R="${GRAFT_REPO_ROOT:-$(pwd)}"
O="$R/gpurun_out/prof_r03"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
run() {  # name, program args...
  local name="$1"; shift
  echo "== $name"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/$name" -- python3 "$@" > "$O/$name.out" 2> "$O/$name.err" || { echo "FAILED $name"; tail -5 "$O/$name.err"; return 1; }
  tail -c 400 "$O/$name.out"; echo
}
STAGE="${1:-all}"    # trace | pmc | all | topk (a gpurun call is limited to 20 minutes: run the stages in separate calls)
if [ "$STAGE" = "topk" ]; then
run topk          "$R/bench.py" --workload topk --steps 3 --warmup 1
exit $?
fi
if [ "$STAGE" != "pmc" ]; then
run gaussian_mf   "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --only &&
run hpf_cavi      "$R/bench.py" --workload hpf_cavi --steps 10 --warmup 2 --no-cpu-baseline &&
run topk          "$R/bench.py" --workload topk --steps 2 --warmup 1 &&
run gaussian_k128 "$R/bench.py" --workload gaussian_mf_k128 --steps 3 --warmup 1 --no-cpu-baseline &&
run ksweep_gauss  "$R/tools/k_sweep.py" gauss 16 30 40 50 72 80 96 100 112 &&
run ksweep_hpf    "$R/tools/k_sweep.py" hpf 16 20 32 40 || exit 1
fi
if [ "$STAGE" != "trace" ]; then
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $c"
  rocprofv3 --pmc $c --output-format csv -d "$R/gpurun_out/pmc_gaussian_mf_$c" -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --only > "$O/pmc_gauss_$c.out" 2> "$O/pmc_gauss_$c.err" || { echo FAILED; tail -5 "$O/pmc_gauss_$c.err"; }
  rocprofv3 --pmc $c --output-format csv -d "$R/gpurun_out/pmc_hpf_cavi_$c" -- python3 "$R/bench.py" --workload hpf_cavi --steps 4 --warmup 1 --no-cpu-baseline > "$O/pmc_hpf_$c.out" 2> "$O/pmc_hpf_$c.err" || { echo FAILED; tail -5 "$O/pmc_hpf_$c.err"; }
  rocprofv3 --pmc $c --output-format csv -d "$R/gpurun_out/pmc_gaussian_mf_k128_$c" -- python3 "$R/bench.py" --workload gaussian_mf_k128 --steps 1 --warmup 1 --no-cpu-baseline > "$O/pmc_k128_$c.out" 2> "$O/pmc_k128_$c.err" || { echo FAILED; tail -5 "$O/pmc_k128_$c.err"; }
done
fi
# keep the merge small: only the stats / counter summaries travel back
find "$R/gpurun_out" -name "*kernel_trace.csv" -delete
find "$R/gpurun_out" -name "*_agent_info.csv" -delete
ls -R "$O" | head -60
