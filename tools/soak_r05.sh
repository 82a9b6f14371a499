#!/bin/bash
# Round-5 soak on the GPU box (through gpurun, from the repo root): the checkers of tests/tools/ with fresh seeds on the round's build, one after the other,
# every one under its own timeout; everything lands in gpurun_out/TAG/soak.txt.      tools/soak_r05.sh TAG
TAG=${1:-soak}
OUT=gpurun_out/$TAG
mkdir -p $OUT
S=$OUT/soak.txt
run() { echo "== $*" >> $S; timeout -k 10 $1 "${@:2}" >> $S 2>&1; echo "   (exit $?)" >> $S; }
run 240 python tests/tools/fuzz_big.py 60000 160
run 120 python tests/tools/fuzz_impact.py 61000 80
run 120 python tests/tools/fuzz_impact.py 62000 80 ap
run 150 python tests/tools/fuzz_lcp.py 63000 250
run 150 python tests/tools/fuzz_lcp.py 64000 60 big
run 150 python tests/tools/fuzz_joints.py 65000 150
run 150 python tests/tools/fuzz_parity.py 6600 6680 300
run 150 python tests/tools/fuzz_artic.py 67000 120
run 100 python tests/tools/fuzz_artic_stab.py 68000 40
run 280 python tests/tools/long_horizon_parity.py 4096 4400
run 200 python tests/tools/full_size_parity.py wheel
grep -v "^case\|^  case" $S | tail -60
