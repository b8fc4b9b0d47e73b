# usage: bash scripts/pmc_icache.sh <tag> [bench args]  -- instruction-cache counters of the dominant kernel of bench.py -> gpurun_out/<tag>_icache.json
tag=${1:-r02}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_ic_a -- python3 bench.py --steps 5 --cpu-sample 0 "$@" > /dev/null 2>&1
rocprofv3 --pmc SQC_ICACHE_BUSY_CYCLES SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_BRANCH GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_ic_b -- python3 bench.py --steps 5 --cpu-sample 0 "$@" > /dev/null 2>&1
python3 - "$tag" <<'PY'
import csv, collections, glob, json, sys
tag = sys.argv[1]
out = {}
for d in ("gpurun_out/pmc_ic_a", "gpurun_out/pmc_ic_b"):
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, c in agg.items():
            if "k_spectrum" in k:
                out.setdefault(k, {}).update({n: sum(v) / len(v) for n, v in c.items()})
json.dump(out, open("gpurun_out/%s_icache.json" % tag, "w"), indent=1)
for k, c in out.items():
    print(k, {n: round(v) for n, v in c.items()})
    if "SQC_ICACHE_REQ" in c: print("   hit rate", c.get("SQC_ICACHE_HITS", 0) / max(c["SQC_ICACHE_REQ"], 1), "misses per wave", c.get("SQC_ICACHE_MISSES", 0) / max(c.get("SQ_WAVES", 1), 1))
PY
