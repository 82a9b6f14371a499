#!/bin/bash
# The headline part of tools/profile_session_r05.sh alone (the round's last build changed the one-wavefront kernels only): bench line at the driver's command, kernel
# stats, FETCH_SIZE / WRITE_SIZE and the two SQ passes of the timed launch.        tools/profile_session_r05_headline.sh TAG
set -e
TAG=${1:-session}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
DRV="--gpus 1 --steps 20 --warmup 5"
CMD="python3 bench.py $DRV --no-cpu-baseline --no-config3 --no-config4 --no-long-horizon --no-config5"
echo "$CMD" > $OUT/command.txt
python3 bench.py $DRV > $OUT/bench.json 2> $OUT/bench.err
echo "bench done" > $OUT/progress.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -- $CMD > $OUT/bench_under_rocprof.json 2> $OUT/ks.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > /dev/null 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > /dev/null 2> $OUT/write.err
echo "traffic done" >> $OUT/progress.txt
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq1 -- $CMD > /dev/null 2> $OUT/sq1.err
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM --output-format csv -d $OUT/sq2 -- $CMD > /dev/null 2> $OUT/sq2.err
echo "sq done" >> $OUT/progress.txt
find $OUT -name "*.db" -delete 2>/dev/null || true
du -sh $OUT
