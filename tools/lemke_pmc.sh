#!/bin/bash
# SQ counter passes of tools/lemke_bench.py for one library variant: tools/lemke_pmc.sh TAG LIB "bench args"
set -e
TAG=$1; LIB=$2; ARGS=$3
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
export MOBY_HIP_LIB=$LIB
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq1 -- python3 tools/lemke_bench.py $ARGS > $OUT/sq1.json 2> $OUT/sq1.err
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM --output-format csv -d $OUT/sq2 -- python3 tools/lemke_bench.py $ARGS > $OUT/sq2.json 2> $OUT/sq2.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 tools/lemke_bench.py $ARGS > /dev/null 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 tools/lemke_bench.py $ARGS > /dev/null 2> $OUT/write.err
find $OUT -name "*.db" -delete 2>/dev/null || true
python3 - <<PY
import csv, glob, collections
for sub in ("sq1","sq2","fetch","write"):
    acc=collections.defaultdict(float); dur=collections.defaultdict(float)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv"%sub, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_lcp_block<1>" not in r["Kernel_Name"]: continue
            acc[r["Counter_Name"]]+=float(r["Counter_Value"])
    print(sub, dict(acc))
PY
