#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes over the K = 128 shard workload (separate runs, counters only).
set -o pipefail
R="${GRAFT_REPO_ROOT:-$(pwd)}"
O="$R/gpurun_out/prof_r02"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $c"
  rocprofv3 --pmc $c --output-format csv -d "$R/gpurun_out/pmc_gaussian_mf_k128_$c" -- python3 "$R/bench.py" --workload gaussian_mf_k128 --steps 1 --warmup 1 --no-cpu-baseline > "$O/pmc_k128_$c.out" 2> "$O/pmc_k128_$c.err" || { echo FAILED; tail -5 "$O/pmc_k128_$c.err"; exit 1; }
done
find "$R/gpurun_out" -name "*_agent_info.csv" -delete
