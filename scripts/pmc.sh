# usage: bash scripts/pmc.sh <tag> [kernel substring] [bench args]  -- SQ counter passes on the bench workload -> profiles/<tag>_sq_counters.json
tag=$1; kern=${2:-k_spectrum_fused<1, 0}; shift; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_a -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 "$@" > /dev/null 2> gpurun_out/pmc_${tag}_a.err
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_b -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 "$@" > /dev/null 2> gpurun_out/pmc_${tag}_b.err
python3 - "$tag" "$kern" <<'PY'
import csv, collections, glob, json, sys
tag, kern = sys.argv[1], sys.argv[2]
out = {"note": "rocprofv3 --pmc, two passes (SQ counters are summed over the device per launch; averages over the launches of the kernel)",
       "kernel_substring": kern, "counters": {}}
for d in ("gpurun_out/pmc_%s_a" % tag, "gpurun_out/pmc_%s_b" % tag):
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                out["kernel"] = r["Kernel_Name"].split("(")[0]
        for k, v in agg.items():
            out["counters"][k] = {"launches": len(v), "avg": sum(v) / len(v)}
c = {k: v["avg"] for k, v in out["counters"].items()}
if "SQ_WAVES" in c and "SQ_INSTS_VALU" in c:
    cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8   # (summed over the 8 XCDs)
    out["derived"] = {"valu_instructions_per_wavefront": c["SQ_INSTS_VALU"] / c["SQ_WAVES"],
                      "valu_busy_cycles_per_simd_4cyc": c.get("SQ_ACTIVE_INST_VALU", 0) * 4 / 1024,
                      "kernel_cycles_per_xcd": cyc}
    if cyc:
        out["derived"]["valu_busy_fraction_4cyc_issue"] = out["derived"]["valu_busy_cycles_per_simd_4cyc"] / cyc
json.dump(out, open("profiles/%s_sq_counters.json" % tag, "w"), indent=1)
print(json.dumps(out, indent=1))
PY
cp profiles/${tag}_sq_counters.json gpurun_out/
