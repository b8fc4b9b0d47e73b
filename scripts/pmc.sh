# usage: bash scripts/pmc.sh <tag>   -- SQ counter passes on the bench workload (k_spectrum<1,1>)
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_a -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 > /dev/null 2> gpurun_out/pmc_${tag}_a.err
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_b -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 > /dev/null 2> gpurun_out/pmc_${tag}_b.err
python3 - <<PY
import csv, collections, glob
for d in ("gpurun_out/pmc_${tag}_a","gpurun_out/pmc_${tag}_b"):
    for f in glob.glob(d+"/*/*counter_collection.csv"):
        rows=list(csv.DictReader(open(f)))
        agg=collections.defaultdict(list)
        for r in rows:
            if "k_spectrum<1, 1" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in agg.items(): print(k, len(v), sum(v)/len(v))
PY
