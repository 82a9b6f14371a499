#!/bin/bash
# The round's closing soak on the GPU box (through gpurun, from the repo root): the checkers of tests/tools/ with fresh seeds on the LAST build, each under its own
# timeout, about ten minutes in all; everything lands in gpurun_out/TAG/soak.txt.      tools/soak_r05_final.sh TAG
TAG=${1:-soakf}
OUT=gpurun_out/$TAG
mkdir -p $OUT
S=$OUT/soak.txt
run() { echo "== $*" >> $S; timeout -k 10 $1 "${@:2}" >> $S 2>&1; echo "   (exit $?)" >> $S; }
run 120 python tests/tools/fuzz_big.py 90000 60
run 60 python tests/tools/fuzz_impact.py 91000 40
run 60 python tests/tools/fuzz_impact.py 92000 40 ap
run 80 python tests/tools/fuzz_lcp.py 93000 120
run 90 python tests/tools/fuzz_lcp.py 94000 30 big
run 80 python tests/tools/fuzz_joints.py 95000 80
run 80 python tests/tools/fuzz_parity.py 9600 9640 300
run 110 python tests/tools/fuzz_throw.py 9700 40
grep -v "^case\|^  case\|^seed \|amdgpu.ids" $S | tail -40
