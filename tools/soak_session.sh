set -e
mkdir -p gpurun_out/soak3
python tests/tools/fuzz_lcp.py 31 120 > gpurun_out/soak3/fuzz_lcp.txt 2>&1; tail -1 gpurun_out/soak3/fuzz_lcp.txt
python tests/tools/fuzz_lcp.py 32 60 big > gpurun_out/soak3/fuzz_lcp_big.txt 2>&1; tail -1 gpurun_out/soak3/fuzz_lcp_big.txt
python tests/tools/fuzz_impact.py 7000 40 > gpurun_out/soak3/fuzz_impact_ds.txt 2>&1; tail -1 gpurun_out/soak3/fuzz_impact_ds.txt
python tests/tools/fuzz_impact.py 7100 30 ap > gpurun_out/soak3/fuzz_impact_ap.txt 2>&1; tail -1 gpurun_out/soak3/fuzz_impact_ap.txt
python tests/tools/fuzz_big.py 7200 40 > gpurun_out/soak3/fuzz_big.txt 2>&1; tail -1 gpurun_out/soak3/fuzz_big.txt
python tests/tools/fuzz_artic_stab.py 7300 120 > gpurun_out/soak3/fuzz_artic_stab.txt 2>&1; tail -1 gpurun_out/soak3/fuzz_artic_stab.txt
python tests/tools/fuzz_artic.py 7400 40 > gpurun_out/soak3/fuzz_artic.txt 2>&1; tail -1 gpurun_out/soak3/fuzz_artic.txt
python tests/tools/fuzz_joints.py 7500 40 > gpurun_out/soak3/fuzz_joints.txt 2>&1; tail -1 gpurun_out/soak3/fuzz_joints.txt
python tests/tools/fuzz_parity.py 7600 7640 200 > gpurun_out/soak3/fuzz_parity.txt 2>&1; tail -1 gpurun_out/soak3/fuzz_parity.txt
