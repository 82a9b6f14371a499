#!/bin/bash
# Round-5 profiling session on the GPU box (through gpurun, from the repo root): gpurun_out/TAG/ in the layout tools/summarise_profiles.py reads.
# The headline passes run the DRIVER's exact command (`bench.py --gpus 1 --steps 20 --warmup 5`; informational legs off under the profiler).
# Counters in their own passes with --kernel-trace only.        tools/profile_session_r05.sh TAG
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
# config 4 at the bench size: three full steps (cold, then warm), then ONE step under the profiler: kernel trace, FETCH_SIZE, WRITE_SIZE
python3 tools/config4_full_size.py 16 1024 3 > $OUT/c4_3steps.json 2> $OUT/c4_3steps.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4_3steps_ks -- python3 tools/config4_full_size.py 16 1024 3 > /dev/null 2> $OUT/c4_3steps_ks.err
echo "config 4 three steps done" >> $OUT/progress.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4step_ks -- python3 tools/config4_full_size.py 16 1024 > $OUT/c4step.json 2> $OUT/c4step_ks.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/c4step_fetch -- python3 tools/config4_full_size.py 16 1024 > /dev/null 2> $OUT/c4step_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/c4step_write -- python3 tools/config4_full_size.py 16 1024 > /dev/null 2> $OUT/c4step_write.err
echo "config 4 done" >> $OUT/progress.txt
# config 4 at BASELINE's size: one full step of 64 boxes x 8 worlds under the kernel trace, and the sizes of the tall-stack leg
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4_64x8_ks -- python3 tools/config4_full_size.py 64 8 1 > $OUT/c4_64x8.json 2> $OUT/c4_64x8.err
python3 tools/config4_full_size.py 32 64 1 > $OUT/c4_32x64.json 2> $OUT/c4_32x64.err
python3 tools/config4_full_size.py 28 64 1 > $OUT/c4_28x64.json 2> $OUT/c4_28x64.err
echo "config 4 tall stacks done" >> $OUT/progress.txt
find $OUT -name "*.db" -delete 2>/dev/null || true
du -sh $OUT
