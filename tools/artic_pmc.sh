#!/bin/bash
# SQ counter passes of the config-5 leg (ur10 x 8192, 200 steps) for both articulated steppers: tools/artic_pmc.sh TAG
set -e
TAG=${1:-artic}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cat > $OUT/run.py <<'PY'
import sys, os, json; sys.path.insert(0, os.getcwd())
import torch, bench
r = bench.config5_leg(torch, cpu=False)
print(json.dumps({k: r[k] for k in ("ms", "world_steps_per_sec", "lcp_rows_per_sec", "worlds_with_errors")}))
PY
for p in 1 0; do
  export MH_ARTIC_PACK=$p
  python3 $OUT/run.py > $OUT/plain_pack$p.json 2> /dev/null
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq1_pack$p -- python3 $OUT/run.py > /dev/null 2> $OUT/sq1_pack$p.err
  rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM --output-format csv -d $OUT/sq2_pack$p -- python3 $OUT/run.py > /dev/null 2> $OUT/sq2_pack$p.err
done
find $OUT -name "*.db" -delete 2>/dev/null || true
python3 - <<PY
import csv, glob, json, collections
out = {"workload": "ur10 x 8192, 200 steps in one launch (bench.py config5_leg), the timed launch", "kernels": {}}
for p, kern in ((1, "k_artic_step_p2"), (0, "k_artic_step_w4")):
    c = {}
    dur = None
    for sub in ("sq1", "sq2"):
        for f in glob.glob("$OUT/%s_pack%d/**/*counter_collection.csv" % (sub, p), recursive=True):
            rows = [r for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"]]
            if not rows: continue
            last = max(int(r["Dispatch_Id"]) for r in rows)
            for r in rows:
                if int(r["Dispatch_Id"]) == last:
                    c[r["Counter_Name"]] = c.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e9
    if not c: continue
    ws = 8192 * 200
    plain = json.load(open("$OUT/plain_pack%d.json" % p))
    out["kernels"][kern] = {"launch_ms_plain": plain["ms"], "kernel_seconds_under_profiler": dur,
        "per_world_step": {"valu_insts": c["SQ_INSTS_VALU"] / ws, "salu_insts": c["SQ_INSTS_SALU"] / ws, "lds_insts": c["SQ_INSTS_LDS"] / ws, "vmem_insts": c["SQ_INSTS_VMEM"] / ws},
        "wave_cycles": {"issuing": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], "waiting_s_waitcnt": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], "issue_stalled": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]},
        "valu_busy_frac": c["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * 2.4e9 * dur)}
print(json.dumps(out, indent=1))
json.dump(out, open("$OUT/artic_issue.json", "w"), indent=1)
PY
