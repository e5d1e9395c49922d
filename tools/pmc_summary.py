"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (gpurun_out/pmc_<workload>_<counter>/)
into profiles/<tag>_hbm_traffic.csv and profiles/traffic.json (bytes per launch of the dominant kernel).

Corrections follow MI355X_MICROARCH.md section HBM: counter unit = 1024 B; on gfx950 FETCH_SIZE
reports half of the bytes of wide (16 B/lane) coalesced reads, so read bytes = 2 x FETCH_SIZE x 1024."""
import collections, csv, glob, json, os, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dominant = {"gaussian_mf": ("gauss_accum", "gauss_accum_mfma_kernel"), "hpf_cavi": ("gamma_sweep", "gamma_sweep_kernel"),
            "gaussian_mf_k128": ("gauss_accum", "gauss_accum_mfma128_kernel")}
rows, traffic = [], {}
for w, (cls, kname) in dominant.items():
    per = collections.defaultdict(dict)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        files = glob.glob(os.path.join(root, "gpurun_out", f"pmc_{w}_{c}", "*", "*_counter_collection.csv"))
        if not files:
            continue
        agg = collections.defaultdict(list)
        newest = max(files, key=os.path.getmtime)
        for r in csv.DictReader(open(newest)):
            if r["Counter_Name"] == c:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            per[k][c] = (sum(v) / len(v), len(v))
    for k, d in per.items():
        if k.startswith("__amd_rocclr") or "rocprim" in k:   # runtime fills / the one-off library sort
            continue
        f, nf = d.get("FETCH_SIZE", (0.0, 0))
        wv, nw = d.get("WRITE_SIZE", (0.0, 0))
        hbm = (2 * f + wv) * 1024
        rows.append([w, k, nf, f, wv, 2 * f * 1024, wv * 1024, hbm])
        if kname in k:
            traffic[f"{w}:{cls}"] = hbm
with open(os.path.join(root, "profiles", f"{tag}_hbm_traffic.csv"), "w", newline="") as fh:
    wr = csv.writer(fh)
    wr.writerow(["workload", "kernel", "launches", "FETCH_SIZE_mean_raw", "WRITE_SIZE_mean_raw",
                 "read_bytes_per_launch(2x corrected)", "write_bytes_per_launch", "hbm_bytes_per_launch"])
    wr.writerows(rows)
json.dump(traffic, open(os.path.join(root, "profiles", "traffic.json"), "w"), indent=1)
print(traffic)
