"""Register / spill / scratch / LDS numbers of every kernel in the shipped code objects (what VERDICT quotes).
usage: python tools/kernel_resources.py [build/obj/mh_capi.o ...]   (default: every object under build/obj)"""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
objs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "build", "obj", "*.o")))
for o in objs:
    with tempfile.TemporaryDirectory() as d:
        tmp = os.path.join(d, os.path.basename(o))
        subprocess.run(["cp", o, tmp]); subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", tmp], capture_output=True, cwd=d)
        cos = glob.glob(tmp + ".*gfx950")
        if not cos:
            continue
        txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", cos[0]], capture_output=True, text=True).stdout
    for blk in txt.split("- .agpr_count:")[1:]:
        g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
        name = g("name")
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
        print("%-28s %-44s vgpr %3s (spill %3s) sgpr %3s (spill %3s) scratch %4s B lds %6s B" % (
            os.path.basename(o), name[:44], g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_count"), g("sgpr_spill_count"),
            g("private_segment_fixed_size"), g("group_segment_fixed_size")))
