#!/bin/bash
# MFMA-pipe busy fraction and the clock the chip holds during the fused top-k (PMC only, no tracing beside it).
set -o pipefail
R="${GRAFT_REPO_ROOT:-$(pwd)}"
O="$R/gpurun_out/pmc_topk"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d "$O/a" -- python3 "$R/bench.py" --workload topk --steps 2 --warmup 1 > "$O/a.out" 2> "$O/a.err" || { echo FAILED a; tail -5 "$O/a.err"; }
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES --output-format csv -d "$O/b" -- python3 "$R/bench.py" --workload topk --steps 2 --warmup 1 > "$O/b.out" 2> "$O/b.err" || { echo FAILED b; tail -5 "$O/b.err"; }
python3 - "$O" <<'PY'
import csv, glob, sys, collections
for d in ('a', 'b'):
    for f in glob.glob(f"{sys.argv[1]}/{d}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for row in csv.DictReader(open(f)):
            kn = row['Kernel_Name'].split('(')[0][:40]
            acc[kn][row['Counter_Name']] += float(row['Counter_Value']); n[(kn, row['Counter_Name'])] += 1
        for kn in acc:
            if 'topk' in kn:
                print(d, kn, {c: (v / n[(kn, c)]) for c, v in acc[kn].items()})
PY
