"""Copy the judged part of a tools/profile.sh run from gpurun_out/ (scratch) into profiles/ (tracked):
   python tools/save_profile.py gpurun_out/prof_r01 r01 [--latest]"""
import sys as _sys
if {"-h", "--help"} & set(_sys.argv[1:]):          # every tool answers --help without touching the GPU
    print(__doc__)
    raise SystemExit(0)
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "summary.txt"), os.path.join(dst, tag + "_rocprofv3_summary.txt"))
shutil.copy(os.path.join(src, "summary.json"), os.path.join(dst, tag + "_rocprofv3_summary.json"))
for name in ("out_kernel_stats.csv",):
    p = os.path.join(src, "trace", name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, tag + "_kernel_stats.csv"))
summ = json.load(open(os.path.join(src, "summary.json")))
if "--latest" in sys.argv:
    best = None
    for k, v in summ.get("traffic", {}).items():
        if "remap_wg_kernel<0, 5, 2" in k or (best is None and ("remap_tile_kernel" in k or "remap_lds_kernel" in k or "remap_wg_kernel" in k)):
            import datetime
            best = dict(v, kernel=k.replace("void dcp::", "").split("(")[0].replace("<0, 5, 2, float>", "<Radial,NF=5,f64lerp>"),
                        rocprof_kernel_name=k, source=tag + "_rocprofv3_summary.json", collected=datetime.date.today().isoformat())
    if best:
        sys.path.insert(0, root)
        import subprocess
        import bench
        dig = os.path.join(src, "kernel_sources.sha256")
        best["kernel_sources_sha256"] = open(dig).read().strip() if os.path.exists(dig) else None
        try:      # the commit this tree is at, and whether its kernel sources are still the profiled ones
            best["git_commit"] = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
            dirty = subprocess.run(["git", "-C", root, "status", "--porcelain", "--", "discorpy_amd/csrc"], capture_output=True, text=True).stdout.strip()
            best["git_commit_holds_the_profiled_sources"] = (not dirty) and best["kernel_sources_sha256"] == bench.kernel_sources_digest()
        except Exception:      # noqa: BLE001
            best["git_commit"] = None
        for k, v in summ.get("traffic", {}).items():       # the stack kernel's pass: config 4, a 256-projection shard, every row
            if "stack_wg_kernel" in k or "stack_lds_kernel" in k:
                best["stack_shard256"] = dict(v, rocprof_kernel_name=k, algorithmic_bytes_per_launch=8 * 256 * 2560 * 2560)
        json.dump(best, open(os.path.join(dst, "pmc_latest.json"), "w"), indent=1)
        print("pmc_latest.json:", best)
print("saved", tag)
