# SQ counters of k_form_factor_2d (128^2 table in LDS, ARTS size)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_2d_a -- python3 scripts/time_2d.py > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_2d_b -- python3 scripts/time_2d.py > /dev/null 2>&1
python3 - <<PY
import csv, collections, glob
for d in ("gpurun_out/pmc_2d_a","gpurun_out/pmc_2d_b"):
    for f in glob.glob(d+"/*/*counter_collection.csv"):
        rows=list(csv.DictReader(open(f)))
        agg=collections.defaultdict(list)
        for r in rows:
            if "k_form_factor_2d<1, true" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in agg.items(): print(k, len(v), sum(v)/len(v))
PY
