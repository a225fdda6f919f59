#!/bin/bash
# A/B of library variants on ONE box: for every discorpy_amd/lib/variants/lib_*.so run "$@" under rocprofv3 --kernel-trace --stats
# and print the kernels' average durations.   tools/ab_variants.sh python tools/time_spline.py --orders 3 --variants 1
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
    rows = [r for r in csv.DictReader(open(f)) if "rocclr" not in r["Name"]]
    print(name, " | ".join("%s %.1f us" % (r["Name"].split("(")[0].replace("void dcp::", "")[:44], float(r["AverageNs"]) / 1e3) for r in rows[:4]))
PY
done
