#!/bin/bash
# Counter passes of the headline kernel at the DRIVER's exact command (`python bench.py --gpus 1 --steps 20 --warmup 5`, informational legs off):
# separate --pmc passes with --kernel-trace only; written under gpurun_out/TAG/ for tools/summarise_profiles.py.   tools/headline_pmc.sh TAG
set -e
TAG=${1:-headline}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-config4 --no-long-horizon --no-config5"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -- $CMD > $OUT/bench_under_rocprof.json 2> $OUT/ks.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > /dev/null 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > /dev/null 2> $OUT/write.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq1 -- $CMD > /dev/null 2> $OUT/sq1.err
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM --output-format csv -d $OUT/sq2 -- $CMD > /dev/null 2> $OUT/sq2.err
find $OUT -name "*.db" -delete 2>/dev/null || true
du -sh $OUT
