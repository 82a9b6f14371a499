#!/bin/bash
# memory / LDS latency counters of tools/lemke_bench.py for one library variant: tools/lemke_pmc3.sh TAG LIB "bench args"
set -e
TAG=$1; LIB=$2; ARGS=$3
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
export MOBY_HIP_LIB=$LIB
rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/lat -- python3 tools/lemke_bench.py $ARGS > $OUT/lat.json 2> $OUT/lat.err
find $OUT -name "*.db" -delete 2>/dev/null || true
python3 - <<PY
import csv, glob, collections
acc=collections.defaultdict(float)
for f in glob.glob("$OUT/lat/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_lcp_block<1>" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]]+=float(r["Counter_Value"])
d=dict(acc); print("$TAG", d)
for k in ("VMEM","LDS","SMEM"):
    if d.get("SQ_INSTS_"+k): print("  avg latency", k, d["SQ_INST_LEVEL_"+k]/d["SQ_INSTS_"+k], "cycles;  in flight per wave-cycle", d["SQ_INST_LEVEL_"+k]/d["SQ_WAVE_CYCLES"])
PY
