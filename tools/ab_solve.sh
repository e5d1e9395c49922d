#!/bin/bash
# A/B of the 64 < K <= 128 row solve: MFMA block sweep (default) against the VALU row / column splits
# (PMF_GAUSS_VALU_SOLVE=1), fused and un-fused, on the C4 shard and on the C2 data at other K.
# Run on the MI355X box from the repository root; one JSON line per run in gpurun_out/ab_solve.jsonl.
R="${GRAFT_REPO_ROOT:-$(pwd)}"
O="$R/gpurun_out/ab_solve.jsonl"
: > "$O"
run() {  # label, env assignment (or -), bench args...
  local label="$1" envs="$2"; shift 2
  local line
  if [ "$envs" = "-" ]; then line=$(python3 "$R/bench.py" "$@" --no-cpu-baseline --only 2>/dev/null | tail -1)
  else line=$(env $envs python3 "$R/bench.py" "$@" --no-cpu-baseline --only 2>/dev/null | tail -1); fi
  python3 - "$label" "$line" >> "$O" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
print(json.dumps({"run": sys.argv[1], "K": d["config"]["n_factors"], "ms_per_step": round(d["ms_per_step"], 2),
                  "frac": round(d["roofline"]["frac"], 4), "kernels_ms": {k: round(v, 2) for k, v in d["kernels_ms_per_step"].items()}}))
PY
  tail -1 "$O"
}
run "k128 shard, mfma solve fused" - --workload gaussian_mf_k128 --steps 3 --warmup 1 &&
run "k128 shard, valu solve fused" PMF_GAUSS_VALU_SOLVE=1 --workload gaussian_mf_k128 --steps 3 --warmup 1 &&
run "k128 shard, mfma solve unfused" PMF_GAUSS_UNFUSED=1 --workload gaussian_mf_k128 --steps 2 --warmup 1 &&
run "k128 shard, valu solve unfused" "PMF_GAUSS_UNFUSED=1 PMF_GAUSS_VALU_SOLVE=1" --workload gaussian_mf_k128 --steps 2 --warmup 1 &&
for K in 72 80 96 100 112; do
  run "C2 data K=$K, mfma solve" - --factors $K --steps 3 --warmup 1 &&
  run "C2 data K=$K, valu solve" PMF_GAUSS_VALU_SOLVE=1 --factors $K --steps 3 --warmup 1 || exit 1
done
