#!/bin/bash
# HBM traffic of the launches of the cubic-spline path on a 4096^2 float32 frame (tools/time_spline.py --orders 3 --variants ${VARIANT:-1};
# VARIANT=6: the three launches of rounds 3-5):
# FETCH_SIZE and WRITE_SIZE in separate rocprofv3 counter passes (counters never together with --stats / trace domains other than
# --kernel-trace), converted as tools/summarize_prof.py does (KB; reads x 2 -- the calibration of tools/calib_copy.hip, which is
# re-run here on kernels that move exactly 1 GiB each way) and set against what each launch moves BY CONSTRUCTION.
#   tools/pmc_spline.sh            (writes gpurun_out/pmc_spline.json + prints a table)
case "${1:-}" in -h|--help) sed -n '2,7p' "$0" | sed 's/^# \{0,1\}//'; exit 0;; esac
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $ROOT/gpurun_out
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/psp_$c /tmp/psc_$c
  (cd $ROOT && timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/psp_$c -o out -- python tools/time_spline.py --orders 3 --variants ${VARIANT:-1} --reps 12 > /tmp/psp_$c.log 2>&1)
done
hipcc --offload-arch=gfx950 -O3 -o /tmp/calib_copy $ROOT/tools/calib_copy.hip > /tmp/psc_build.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/psc_$c -o out -- /tmp/calib_copy > /tmp/psc_$c.log 2>&1
done
python - $ROOT <<'PY'
import csv, glob, sys, collections, json, os
root = sys.argv[1]
def collect(prefix, want):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob("/tmp/%s_%s/**/out_counter_collection.csv" % (prefix, c), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].split("(")[0].replace("void dcp::", "").replace("void ", "")
                if want(k):
                    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc
cal = collect("psc", lambda k: True)
scale_r, scale_w, calib = 2.0, 1.0, {}
for k, v in cal.items():
    if v["FETCH_SIZE"] and v["WRITE_SIZE"]:
        rd, wr = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"]) * 1024.0, sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"]) * 1024.0
        calib[k] = {"FETCH_SIZE_KB_as_bytes": rd, "WRITE_SIZE_KB_as_bytes": wr, "moved_each_way": float(1 << 30)}
        print("calibration %-40s FETCH_SIZE x 1024 = %.4f GiB, WRITE_SIZE x 1024 = %.4f GiB for 1 GiB each way" % (k[:40], rd / (1 << 30), wr / (1 << 30)))
acc = collect("psp", lambda k: "spline" in k)
H = W = 4096
P = H * W
# bytes each launch moves by construction (float32 frame 4 B / px, float64 coefficient plane 8 B / px)
by_design = {"spline_col": (4 * P, 8 * P), "spline_row": (8 * P, 8 * P), "spline_wg": (8 * P, 4 * P), "spline_prefilter2d": (4 * P, 8 * P)}
out = {"frame": [H, W], "order": 3, "kernels": {}, "calibration": calib,
       "conversion": "FETCH_SIZE / WRITE_SIZE are in KB; reads x 2 on gfx950 (MI355X_MICROARCH.md, HBM / rocprofv3 section; checked above)"}
tot_r = tot_w = 0.0
for k, v in sorted(acc.items()):
    rd = sum(v["FETCH_SIZE"]) / max(1, len(v["FETCH_SIZE"])) * 1024.0 * scale_r
    wr = sum(v["WRITE_SIZE"]) / max(1, len(v["WRITE_SIZE"])) * 1024.0 * scale_w
    fam = next((f for f in by_design if k.startswith(f)), None)
    d = by_design.get(fam, (0, 0))
    tot_r += rd
    tot_w += wr
    out["kernels"][k] = {"read_bytes": rd, "written_bytes": wr, "by_design_read": d[0], "by_design_written": d[1], "launches": len(v["FETCH_SIZE"])}
    print("%-56s read %7.1f MB (design %6.1f)  written %7.1f MB (design %6.1f)  n=%d" % (k[:56], rd / 1e6, d[0] / 1e6, wr / 1e6, d[1] / 1e6, len(v["FETCH_SIZE"])))
out["total_bytes_per_frame"] = tot_r + tot_w
out["algorithmic_bytes_per_frame"] = 8 * P
design = sum(sum(by_design[f]) for f in by_design if any(k.startswith(f) for k in acc))
out["by_design_bytes_per_frame"] = design
print("per frame: %.1f MB measured, %.1f MB by design (%d B / px), %.1f MB algorithmic (8 B / px): %.2f x algorithmic" % (
    (tot_r + tot_w) / 1e6, design / 1e6, design // P, 8 * P / 1e6, (tot_r + tot_w) / (8 * P)))
json.dump(out, open(os.path.join(root, "gpurun_out", "pmc_spline.json"), "w"), indent=1)
# (copy to profiles/pmc_spline_latest.json to have bench.py quote it as the cubic entry's `traffic`)
PY
