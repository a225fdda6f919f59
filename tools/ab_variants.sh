#!/bin/bash
# A/B of library variants on ONE box: for every discorpy_amd/lib/variants/lib_*.so run "$@" under rocprofv3 --kernel-trace --stats
# and print the kernels' average durations.   tools/ab_variants.sh python tools/time_spline.py --orders 3 --variants 1
case "${1:-}" in -h|--help) sed -n '2,3p' "$0" | sed 's/^# \{0,1\}//'; exit 0;; esac
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
for lib in $ROOT/discorpy_amd/lib/variants/lib_*.so; do
  name=$(basename $lib .so)
  rm -rf /tmp/abv_$name
  (cd /tmp && DCP_LIB_PATH=$lib rocprofv3 --kernel-trace --stats -d /tmp/abv_$name -o out --output-format csv -- "$@" > /tmp/abv_$name.log 2>&1)
  python - "$name" <<'PY'
import csv, glob, sys
name = sys.argv[1]
for f in glob.glob("/tmp/abv_%s/**/out_kernel_stats.csv" % name, recursive=True):
    import os
    rows = [r for r in csv.DictReader(open(f)) if "rocclr" not in r["Name"]]
    want = [w for w in os.environ.get("AB_KERNELS", "").split(",") if w]        # AB_KERNELS=remap_lds_kernel<0, 9,stack_wg: only these
    rows = [r for r in rows if any(w in r["Name"] for w in want)] if want else rows[:4]
    print(name, " | ".join("%s %.1f us" % (r["Name"].split("(")[0].replace("void dcp::", "")[:44], float(r["AverageNs"]) / 1e3) for r in rows))
PY
done
