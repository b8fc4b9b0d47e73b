# usage: bash scripts/pmc_2d.sh <tag>  -- SQ counters of k_form_factor_2d (128^2 table in LDS, ARTS size) -> gpurun_out/<tag>_2d_sq_counters.json
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_2d_a -- python3 scripts/time_2d.py > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_2d_b -- python3 scripts/time_2d.py > /dev/null 2>&1
python3 - "$tag" <<'PY'
import csv, collections, glob, json, sys
tag = sys.argv[1]
out = {"note": "rocprofv3 --pmc, two passes over scripts/time_2d.py; averages over the launches of each kernel (counters summed over the device)", "kernels": {}}
for kern in ("k_form_factor_2d<1, true, 4, false>", "k_form_factor_2d<1, true, 4, true>", "k_form_factor_2d<1, false, 1, false>", "k_form_factor_2d<1, false, 1, true>",
             "k_form_factor_2d_adj<1, false, 1>", "k_form_factor_2d_adj<1, true, 4>", "k_ff2d_table_adj"):
    c = {}
    for d in ("gpurun_out/pmc_2d_a", "gpurun_out/pmc_2d_b"):
        for f in glob.glob(d + "/*/*counter_collection.csv"):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if kern in r["Kernel_Name"]:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, v in agg.items():
                c[k] = sum(v) / len(v)
    if not c:
        continue
    cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8
    der = {}
    if cyc:
        der = {"kernel_cycles_per_xcd": cyc, "valu_busy_fraction_4cyc": c.get("SQ_ACTIVE_INST_VALU", 0) * 4 / 1024 / cyc,
               "lds_busy_fraction (SQ_ACTIVE_INST_LDS x 4 / 1024 SIMDs)": c.get("SQ_ACTIVE_INST_LDS", 0) * 4 / 1024 / cyc,
               "lds_bank_conflict_cycles_per_cu_over_kernel": c.get("SQ_LDS_BANK_CONFLICT", 0) / 256 / cyc,
               "lds_idx_active_cycles_per_cu_over_kernel": c.get("SQ_LDS_IDX_ACTIVE", 0) / 256 / cyc,
               "valu_instructions_per_wavefront": c.get("SQ_INSTS_VALU", 0) / max(c.get("SQ_WAVES", 1), 1),
               "lds_instructions_per_wavefront": c.get("SQ_INSTS_LDS", 0) / max(c.get("SQ_WAVES", 1), 1)}
    out["kernels"][kern] = {"counters": c, "derived": der}
json.dump(out, open("gpurun_out/%s_2d_sq_counters.json" % tag, "w"), indent=1)
print(json.dumps({k: v["derived"] for k, v in out["kernels"].items()}, indent=1))
PY
