"""Cut the judged summaries out of one profiling session (the directory layout the command in profiles/README.md writes:
<dir>/bench.json, bench_under_rocprof.json, ks/ (rocprofv3 --kernel-trace --stats), fetch/, write/ (FETCH_SIZE / WRITE_SIZE passes),
sq1/, sq2/ (the two SQ_* passes)) into profiles/<tag>_*: python tools/summarise_profiles.py gpurun_out/r02d r02_d"""
import csv, glob, json, os, shutil, sys

src, tag = sys.argv[1], sys.argv[2]
import subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMIT = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None   # the build the session ran (commit before profiling)
P = os.path.join(ROOT, "profiles")
CMDTXT = open(os.path.join(src, "command.txt")).read().strip() if os.path.exists(os.path.join(src, "command.txt")) else "python3 bench.py --no-cpu-baseline --no-config4 --no-long-horizon --no-config5"
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
         "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE --output-format csv -- %s (separate passes)" % CMDTXT,
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
    out = {"commit": COMMIT, "source": "profiles/%s_world_step_sq_pmc.csv (two rocprofv3 --pmc passes of `%s`, the timed launch)" % (tag, CMDTXT),
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


# ---- config 4: the block solver's kernels in a cold impact-handler call and in one full step at 16 boxes x 1024 worlds ----
import collections


def per_kernel(csvfile, col):
    tot = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(csvfile)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        tot[k] += float(r[col]) if col != "dur" else (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e9
        n[k] += 1
    return tot, n


if os.path.exists(os.path.join(src, "c4_impact_bench.jsonl")):
    shutil.copy(os.path.join(src, "c4_impact_bench.jsonl"), os.path.join(P, tag + "_config4_impact_bench.jsonl"))
ks4 = one("c4ks/*/*_kernel_stats.csv")
if ks4:
    csv.writer(open(os.path.join(P, tag + "_config4_impact_kernel_stats.csv"), "w")).writerows(list(csv.reader(open(ks4)))[:8])
if os.path.exists(os.path.join(src, "c4step.json")):
    line = [l for l in open(os.path.join(src, "c4step.json")) if l.startswith("{")]
    if line:
        open(os.path.join(P, tag + "_config4_16x1024.json"), "w").write(line[-1])
kt4 = one("c4step_ks/*/*_kernel_trace.csv")
if kt4:
    rd = csv.reader(open(kt4)); h = next(rd)
    rows = [(r[h.index("Kernel_Name")].split("(")[0].replace("void ", ""), int(r[h.index("Start_Timestamp")]), int(r[h.index("End_Timestamp")])) for r in rd if "k_lcp_block" in r[h.index("Kernel_Name")]]
    if rows:
        t0 = rows[0][1]; tot = collections.defaultdict(float)
        out = ["# rocprofv3 --kernel-trace --stats -- python3 tools/config4_full_size.py 16 1024 (commit %s): the block solver's launches of ONE full step" % COMMIT,
               "# kernel, start [s], duration [s]   (k_lcp_block<0>: the lcp_fast kinds; <1>: the Lemke ladder's (world, attempt) tasks)"]
        for k, a, b in rows:
            out.append("%-32s %8.3f %8.3f" % (k, (a - t0) / 1e9, (b - a) / 1e9)); tot[k] += (b - a) / 1e9
        out.append("# totals: " + ", ".join("%s %.1f s" % kv for kv in tot.items()))
        open(os.path.join(P, tag + "_config4_step_kernel_trace.txt"), "w").write("\n".join(out) + "\n")
        print(out[-1])
f4, w4 = one("c4step_fetch/*/*_counter_collection.csv"), one("c4step_write/*/*_counter_collection.csv")
if f4 and w4:
    res = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (then WRITE_SIZE, separate passes) --output-format csv -- python3 tools/config4_full_size.py 16 1024",
           "workload": "ONE full step of 16-box stacks (impact LCP n = 512) x 1024 worlds", "commit": COMMIT,
           "units": "counter value x 1 KiB, summed over the kernel's launches of the step", "kernels": {}}
    for kind, f in (("fetch", f4), ("write", w4)):
        tot, n = per_kernel(f, "Counter_Value")
        for k, v in tot.items():
            if v * 1024 / 1e9 > 0.05:
                res["kernels"].setdefault(k, {"launches": n[k]})[kind + "_GB"] = round(v * 1024 / 1e9, 2)
    dur, _ = per_kernel(one("c4step_fetch/*/*_kernel_trace.csv"), "dur")
    for k in res["kernels"]:
        res["kernels"][k]["seconds"] = round(dur.get(k, 0.0), 3)
    json.dump(res, open(os.path.join(P, tag + "_config4_step_pmc.json"), "w"), indent=1)
    for k, v in res["kernels"].items():
        if "k_lcp_block" in k:
            print(k, v)

kt3 = one("c4_3steps_ks/*/*_kernel_trace.csv")
if kt3:
    rd = csv.reader(open(kt3)); h = next(rd)
    rows = sorted([(int(r[h.index("Start_Timestamp")]), int(r[h.index("End_Timestamp")]), r[h.index("Kernel_Name")].split("(")[0].replace("void ", "")) for r in rd])
    blk = [r for r in rows if "k_lcp_block" in r[2]]
    if blk:
        # the steps: gaps of more than 0.2 s between launches of the block solver do not occur inside a step; cut at the three largest... simpler: cut where a
        # download (no kernel at all for > 50 ms) separates steps
        t0 = rows[0][0]; cuts = [rows[0][0]]
        for a, b in zip(rows, rows[1:]):
            if b[0] - a[1] > 50e6: cuts.append(b[0])
        cuts.append(rows[-1][1])
        out = ["# rocprofv3 --kernel-trace -- python3 tools/config4_full_size.py 16 1024 3 (commit %s): seconds per kernel family in every stretch of launches" % COMMIT,
               "# (stretches are separated by host-side gaps > 50 ms: the downloads between steps)"]
        for k in range(len(cuts) - 1):
            tot = collections.defaultdict(float)
            for a, b, nme in rows:
                if cuts[k] <= a < cuts[k + 1]: tot[nme] += (b - a) / 1e9
            top = sorted(tot.items(), key=lambda kv: -kv[1])[:4]
            out.append("stretch %d: %.2f s .. %.2f s: " % (k, (cuts[k] - t0) / 1e9, (cuts[k + 1] - t0) / 1e9) + ", ".join("%s %.2f s" % kv for kv in top))
        open(os.path.join(P, tag + "_config4_3steps_kernel_trace.txt"), "w").write("\n".join(out) + "\n")
        print("\n".join(out[2:]))
if os.path.exists(os.path.join(src, "c4_3steps.json")):
    line = [l for l in open(os.path.join(src, "c4_3steps.json")) if l.startswith("{")]
    if line:
        open(os.path.join(P, tag + "_config4_16x1024_3steps.json"), "w").write(line[-1])
