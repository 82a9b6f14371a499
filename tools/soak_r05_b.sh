OUT=gpurun_out/r5o; mkdir -p $OUT; S=$OUT/soak2.txt
run() { echo "== $*" >> $S; timeout -k 10 $1 "${@:2}" >> $S 2>&1; echo "   (exit $?)" >> $S; }
run 420 python tests/tools/fuzz_big.py 70000 150
run 300 python tests/tools/fuzz_artic.py 71000 60
run 300 python tests/tools/fuzz_parity.py 7200 7260 300
grep -v "^seed \|^case\|^  case\|^\[" $S | tail -20
