"""Copy the round's rocprofv3 summaries from gpurun_out/prof_<tag>/ (scratch) into profiles/ (tracked):
kernel-stats CSVs (device kernels only, library / copy kernels dropped), the bench lines printed under the
profiler, and the per-K sweep lines.    python tools/collect_profiles.py r03"""
import csv
import glob
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
names = {"gaussian_mf": "gaussian_mf", "hpf_cavi": "hpf_cavi", "topk": "topk", "gaussian_k128": "gaussian_mf_k128",
         "ksweep_gauss": "ksweep_gaussian_mf", "ksweep_hpf": "ksweep_hpf_cavi"}
for run, name in names.items():
    files = glob.glob(os.path.join(src, run, "*", "*_kernel_stats.csv"))
    if not files:
        print("missing", run)
        continue
    rows = list(csv.reader(open(max(files, key=os.path.getmtime))))
    keep = [rows[0]] + [r for r in rows[1:] if not r[0].startswith("__amd_rocclr") and "rocprim" not in r[0]]
    with open(os.path.join(dst, f"{tag}_{name}_kernel_stats.csv"), "w", newline="") as fh:
        csv.writer(fh, quoting=csv.QUOTE_NONNUMERIC).writerows(keep)
    out = os.path.join(src, f"{run}.out")
    if os.path.exists(out):
        lines = [l for l in open(out) if l.startswith("{")]
        ext = "jsonl" if run.startswith("ksweep") else "json"
        with open(os.path.join(dst, f"{tag}_{name}_bench_under_rocprof.{ext}"), "w") as fh:
            fh.writelines(lines)
    print("wrote", name, len(keep) - 1, "kernels")
