#!/bin/bash
# usage: bash scripts/pmc_cfg4.sh <tag>  -- SQ / TA / TCP counters of the configs[3] sampler (k_form_factor_2d<1, false, 1, false>, bench.py --config4)
# -> gpurun_out/<tag>_cfg4_counters.json.  Separate --pmc passes (no trace domains beside --kernel-trace).
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc_cfg4_$n -- python3 bench.py --config4 --cpu-sample 0 --steps 3 --warmup 1 > /dev/null 2> gpurun_out/pmc_cfg4_$n.err; }
n=a; run SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
n=b; run SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU
# (TA_* / TCP_* counters are NOT collected: a pass with TA_TA_BUSY_sum aborted inside rocprofv3 (signal 6) and left the run hanging until
#  the silence guard killed it -- r03m)
python3 - "$tag" <<'PY'
import csv, collections, glob, json, sys
tag = sys.argv[1]
kern = "k_form_factor_2d<1, false, 1, false>"
c = {}
for d in sorted(glob.glob("gpurun_out/pmc_cfg4_?")):
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            c[k] = sum(v) / len(v)
cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8
der = {}
if cyc:
    waves = max(c.get("SQ_WAVES", 1), 1)
    der = {"kernel_cycles_per_xcd": cyc,
           "valu_busy_fraction_4cyc": c.get("SQ_ACTIVE_INST_VALU", 0) * 4 / 1024 / cyc,
           "valu_instructions_per_wavefront": c.get("SQ_INSTS_VALU", 0) / waves,
           "vmem_instructions_per_wavefront": c.get("SQ_INSTS_VMEM", 0) / waves,
           "wait_inst_any_fraction_of_wave_cycles": c.get("SQ_WAIT_INST_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1),
           "valu_instructions_per_sample": c.get("SQ_INSTS_VALU", 0) / (1024.0 * 512 * 256 * 256 / 64),
           "vmem_instructions_per_sample": c.get("SQ_INSTS_VMEM", 0) / (1024.0 * 512 * 256 * 256 / 64)}
json.dump({"note": "rocprofv3 --pmc passes over bench.py --config4; averages over the launches of the kernel (counters summed over the device)", "kernel": kern, "counters": c, "derived": der},
          open("gpurun_out/%s_cfg4_counters.json" % tag, "w"), indent=1)
print(json.dumps({"counters": c, "derived": der}, indent=1))
PY
