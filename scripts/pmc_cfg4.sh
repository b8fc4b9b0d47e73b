#!/bin/bash
# usage: bash scripts/pmc_cfg4.sh <tag>  -- SQ / TA / TCP counters of the configs[3] sampler (k_form_factor_2d<1, false, 1, false>, bench.py --config4)
# -> gpurun_out/<tag>_cfg4_counters.json.  Separate --pmc passes (no trace domains beside --kernel-trace).
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc_cfg4_$n -- python3 bench.py --config4 --cpu-sample 0 --steps 3 --warmup 1 > /dev/null 2> gpurun_out/pmc_cfg4_$n.err; }
n=a; run SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
n=b; run SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU
n=c; run TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum
n=d; run TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
n=e; run TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum
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
           "ta_busy_fraction (TA_TA_BUSY_sum / 256 TAs / cycles)": c.get("TA_TA_BUSY_sum", 0) / 256 / cyc,
           "tcp_accesses_per_vmem_instruction": c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / max(c.get("SQ_INSTS_VMEM", 1), 1),
           "l1_miss_fraction (TCC read requests / cache accesses)": c.get("TCP_TCC_READ_REQ_sum", 0) / max(c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 1), 1),
           "vmem_level_avg_per_simd (SQ_INST_LEVEL_VMEM / busy cycles / 1024... see counters)": c.get("SQ_INST_LEVEL_VMEM", 0) / max(c.get("SQ_BUSY_CYCLES", 1), 1)}
json.dump({"note": "rocprofv3 --pmc passes over bench.py --config4; averages over the launches of the kernel (counters summed over the device)", "kernel": kern, "counters": c, "derived": der},
          open("gpurun_out/%s_cfg4_counters.json" % tag, "w"), indent=1)
print(json.dumps({"counters": c, "derived": der}, indent=1))
PY
