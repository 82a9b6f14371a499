#!/bin/bash
# I-cache / LDS / branch counter pass of tools/lemke_bench.py for one library variant: tools/lemke_pmc2.sh TAG LIB "bench args"
set -e
TAG=$1; LIB=$2; ARGS=$3
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
export MOBY_HIP_LIB=$LIB
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $OUT/ic -- python3 tools/lemke_bench.py $ARGS > $OUT/ic.json 2> $OUT/ic.err
find $OUT -name "*.db" -delete 2>/dev/null || true
python3 - <<PY
import csv, glob, collections
acc=collections.defaultdict(float)
for f in glob.glob("$OUT/ic/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_lcp_block<1>" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]]+=float(r["Counter_Value"])
print("$TAG", dict(acc))
PY
