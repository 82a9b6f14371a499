#!/bin/bash
# One profiling session on the GPU box (run through gpurun from the repo root): writes gpurun_out/<tag>/ in the layout
# tools/summarise_profiles.py cuts the judged summaries from.   tools/profile_session.sh TAG
# Counters are collected in their own passes with --kernel-trace only (never with the trace domains gpurun refuses).
set -e
TAG=${1:-session}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 bench.py --no-cpu-baseline --no-config4 --no-long-horizon --no-config5"
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench done" > $OUT/progress.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -- python3 bench.py --no-cpu-baseline --no-config4 --no-long-horizon > $OUT/bench_under_rocprof.json 2> $OUT/ks.err
echo "ks done" >> $OUT/progress.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > /dev/null 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > /dev/null 2> $OUT/write.err
echo "traffic done" >> $OUT/progress.txt
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq1 -- $CMD > /dev/null 2> $OUT/sq1.err
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM --output-format csv -d $OUT/sq2 -- $CMD > /dev/null 2> $OUT/sq2.err
echo "sq done" >> $OUT/progress.txt
# config 4: the block solver's kernels in one cold impact-handler call (16 boxes x 256 worlds) and their HBM traffic (8 boxes x 256)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4ks -- python3 tools/impact_bench.py 16:256:0 > $OUT/c4_impact_bench.jsonl 2> $OUT/c4ks.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/c4fetch -- python3 tools/impact_bench.py 8:256:0 > /dev/null 2> $OUT/c4fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/c4write -- python3 tools/impact_bench.py 8:256:0 > $OUT/c4_impact_bench_8.jsonl 2> $OUT/c4write.err
echo "config 4 impact done" >> $OUT/progress.txt
# config 4 at full size: ONE full step of 16 boxes x 1024 worlds -- where its time goes (kernel trace) and what it moves through HBM
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4step_ks -- python3 tools/config4_full_size.py 16 1024 > $OUT/c4step.json 2> $OUT/c4step_ks.err
echo "config 4 step trace done" >> $OUT/progress.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/c4step_fetch -- python3 tools/config4_full_size.py 16 1024 > /dev/null 2> $OUT/c4step_fetch.err
echo "config 4 step fetch done" >> $OUT/progress.txt
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/c4step_write -- python3 tools/config4_full_size.py 16 1024 > /dev/null 2> $OUT/c4step_write.err
echo "config 4 done" >> $OUT/progress.txt
# keep what is merged back small: drop everything but the csv summaries
find $OUT -name "*.db" -delete 2>/dev/null || true
du -sh $OUT
