set -e
# the fuzzers of tests/tools against the oracle, on the GPU box (run through gpurun from the repo root); every tool prints as it goes
O=gpurun_out/soak5
mkdir -p $O
python tests/tools/fuzz_lcp.py 51 80 > $O/fuzz_lcp.txt 2>&1; tail -1 $O/fuzz_lcp.txt
python tests/tools/fuzz_lcp.py 52 40 big > $O/fuzz_lcp_big.txt 2>&1; tail -1 $O/fuzz_lcp_big.txt
python tests/tools/fuzz_impact.py 9000 30 > $O/fuzz_impact_ds.txt 2>&1; tail -1 $O/fuzz_impact_ds.txt
python tests/tools/fuzz_big.py 9200 30 > $O/fuzz_big.txt 2>&1; tail -1 $O/fuzz_big.txt
python tests/tools/fuzz_artic.py 9400 16 > $O/fuzz_artic.txt 2>&1; tail -1 $O/fuzz_artic.txt
python tests/tools/fuzz_joints.py 9500 40 > $O/fuzz_joints.txt 2>&1; tail -1 $O/fuzz_joints.txt
python tests/tools/fuzz_parity.py 9600 9640 200 > $O/fuzz_parity.txt 2>&1; tail -1 $O/fuzz_parity.txt
