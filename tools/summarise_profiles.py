"""Cut the judged summaries out of one profiling session (the directory layout the command in profiles/README.md writes:
<dir>/bench.json, bench_under_rocprof.json, ks/ (rocprofv3 --kernel-trace --stats), fetch/, write/ (FETCH_SIZE / WRITE_SIZE passes),
sq1/, sq2/ (the two SQ_* passes)) into profiles/<tag>_*: python tools/summarise_profiles.py gpurun_out/r02d r02_d"""
import csv, glob, json, os, shutil, sys

src, tag = sys.argv[1], sys.argv[2]
import subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMIT = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None   # the build the session ran (commit before profiling)
P = os.path.join(ROOT, "profiles")
one = lambda pat: (glob.glob(os.path.join(src, pat)) or [None])[0]
shutil.copy(os.path.join(src, "bench.json"), os.path.join(P, tag + "_bench.json"))
if os.path.exists(os.path.join(src, "bench_under_rocprof.json")):
    shutil.copy(os.path.join(src, "bench_under_rocprof.json"), os.path.join(P, tag + "_bench_under_rocprof.json"))
ks = one("ks/*/*_kernel_stats.csv")
if ks:
    rows = list(csv.reader(open(ks)))
    csv.writer(open(os.path.join(P, tag + "_kernel_stats.csv"), "w")).writerows(rows[:6])
kt = one("ks/*/*_kernel_trace.csv")
if kt:
    rd = csv.reader(open(kt)); h = next(rd)
    rows = [r for r in rd if "mh_k_world_step" in r[h.index("Kernel_Name")] or "k_artic_step" in r[h.index("Kernel_Name")]]
    w = csv.writer(open(os.path.join(P, tag + "_kernel_trace_rows.csv"), "w")); w.writerow(h); w.writerows(rows)
    for r in rows:
        print("%-44s %.3f ms" % (r[h.index("Kernel_Name")][:44], (int(r[h.index("End_Timestamp")]) - int(r[h.index("Start_Timestamp")])) / 1e6))


def counters(dirs):
    out, hdr = [], None
    for d in dirs:
        f = one(d + "/*/*_counter_collection.csv")
        if not f:
            continue
        rd = csv.reader(open(f)); hdr = next(rd)
        out += [r for r in rd if "mh_k_world_step" in r[hdr.index("Kernel_Name")]]
    return hdr, out


def value(hdr, rows, name):          # the LAST dispatch of the kernel = the timed launch
    sel = [r for r in rows if r[hdr.index("Counter_Name")] == name]
    last = max(int(r[hdr.index("Dispatch_Id")]) for r in sel)
    r = [r for r in sel if int(r[hdr.index("Dispatch_Id")]) == last][0]
    return float(r[hdr.index("Counter_Value")]), (int(r[hdr.index("End_Timestamp")]) - int(r[hdr.index("Start_Timestamp")])) / 1e9


bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
B, steps = bench["config"]["worlds_per_gpu"], bench["steps"]
hdr, rows = counters(["fetch", "write"])
if rows:
    w = csv.writer(open(os.path.join(P, tag + "_world_step_pmc.csv"), "w")); w.writerow(hdr); w.writerows(rows)
    t = {"source": "profiles/%s_world_step_pmc.csv" % tag, "commit": COMMIT,
         "command": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE --output-format csv -- python3 bench.py --no-cpu-baseline --no-config4 --no-long-horizon --no-config5 (separate passes)",
         "kernel": "mh::small::mh_k_world_step", "worlds": B, "steps": steps,
         "fetch_size_kb": value(hdr, rows, "FETCH_SIZE")[0], "write_size_kb": value(hdr, rows, "WRITE_SIZE")[0],
         "note": "the timed launch of each pass; dword-per-lane scratch accesses, uncalibrated width (guide: FETCH_SIZE may under-report up to 2x); bench.py normalises per world-step"}
    json.dump(t, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
    print("traffic: fetch %.3f GB write %.3f GB per launch" % (t["fetch_size_kb"] * 1024 / 1e9, t["write_size_kb"] * 1024 / 1e9))
hdr, rows = counters(["sq1", "sq2"])
if rows:
    w = csv.writer(open(os.path.join(P, tag + "_world_step_sq_pmc.csv"), "w")); w.writerow(hdr); w.writerows(rows)
    c = {n: value(hdr, rows, n)[0] for n in set(r[hdr.index("Counter_Name")] for r in rows)}
    dur = value(hdr, rows, "SQ_ACTIVE_INST_VALU")[1]
    simds, clk = 256 * 4, 2.4e9
    out = {"commit": COMMIT, "source": "profiles/%s_world_step_sq_pmc.csv (two rocprofv3 --pmc passes of `python3 bench.py --no-cpu-baseline --no-config4 --no-long-horizon --no-config5`, the timed launch)" % tag,
           "kernel": "mh::small::mh_k_world_step", "worlds": B, "steps": steps, "kernel_seconds": dur,
           "per_world_step": {"valu_insts": c["SQ_INSTS_VALU"] / (B * steps), "salu_insts": c["SQ_INSTS_SALU"] / (B * steps),
                              "lds_insts": c["SQ_INSTS_LDS"] / (B * steps), "vmem_insts": c["SQ_INSTS_VMEM"] / (B * steps)},
           "wave_cycles": {"issuing": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], "waiting_s_waitcnt": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
                           "issue_stalled": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]},
           "valu_busy_frac": c["SQ_ACTIVE_INST_VALU"] * 4 / (simds * clk * dur),
           "note": "valu_busy_frac = SQ_ACTIVE_INST_VALU (quad-cycles) x 4 / (1024 SIMDs x 2.4 GHz x kernel time): the share of SIMD cycles with the vector ALU occupied -- the kernel's real roofline (issue), next to the HBM byte model"}
    json.dump(out, open(os.path.join(P, "pmc_issue.json"), "w"), indent=1)
    print("valu busy %.3f; per world-step: %s" % (out["valu_busy_frac"], out["per_world_step"]))
print("value %.4g %s, kernel %.1f us, frac %.4f" % (bench["value"], bench["unit"], bench["roofline"]["kernel_avg_us"], bench["roofline"]["frac"]))
