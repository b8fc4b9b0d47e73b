"""profiles/<tag>_traffic.json from the two rocprofv3 --pmc passes of scripts/profile_round.sh (FETCH_SIZE, WRITE_SIZE).
usage: python scripts/traffic.py <tag> <B> <ppp> [kernel name substring, default the loss + gradient kernels of bench.py]"""
import collections, csv, glob, json, sys

tag, B, ppp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
KEYS = (sys.argv[4],) if len(sys.argv) > 4 else ("k_spectrum_fused<1, 0", "k_spectrum<1, 1")
out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (bench.py --steps 5), KB per launch of the dominant "
               "kernel; gfx950: FETCH_SIZE is doubled before use (MI355X_MICROARCH.md, HBM section)",
       "B": B, "ppp": ppp}
for name, d in (("FETCH_SIZE", f"gpurun_out/pmcf_{tag}"), ("WRITE_SIZE", f"gpurun_out/pmcw_{tag}")):
    vals = collections.defaultdict(list)
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and any(k in r["Kernel_Name"] for k in KEYS):
                vals[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    k = max(vals, key=lambda n: len(vals[n]))
    v = vals[k][1:] if len(vals[k]) > 1 else vals[k]  # drop the warm-up launch
    out["kernel"] = k.split("(")[0]
    out[name + "_KB_avg"] = sum(v) / len(v)
    out[name + "_launches"] = len(v)
out["hbm_bytes_per_launch"] = 1024.0 * (2.0 * out["FETCH_SIZE_KB_avg"] + out["WRITE_SIZE_KB_avg"])
json.dump(out, open(f"profiles/{tag}_traffic.json", "w"), indent=1)
print(json.dumps(out))
