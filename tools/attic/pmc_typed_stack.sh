#!/bin/bash
# HBM traffic of the stack kernels by element type (tools/time_typed_stack.py, one launch per case): FETCH_SIZE and WRITE_SIZE in
# separate rocprofv3 counter passes, converted as tools/summarize_prof.py does (KB; reads x 2 -- the calibration of tools/calib_copy.hip)
# and set against the algorithmic bytes of the shard (2 x element size per voxel).   tools/pmc_typed_stack.sh [depth]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
D=${1:-64}
export TMPDIR=/tmp TTS_QUICK=1
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pts_$c
  (cd $ROOT && timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pts_$c -o out -- python tools/time_typed_stack.py $D > /tmp/pts_$c.log 2>&1)
done
python - $D <<'PY'
import csv, glob, sys, collections
D = int(sys.argv[1])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("/tmp/pts_%s/**/out_counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void dcp::", "")
            if "stack" in k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
es = {"float": 4, "double": 8, "unsigned short": 2, "short": 2, "int": 4, "unsigned int": 4}
for k, v in sorted(acc.items()):
    rd = sum(v["FETCH_SIZE"]) / max(1, len(v["FETCH_SIZE"])) * 1024.0 * 2.0
    wr = sum(v["WRITE_SIZE"]) / max(1, len(v["WRITE_SIZE"])) * 1024.0
    print("%-64s read %8.1f MB  written %8.1f MB per launch (n=%d)" % (k[:64], rd / 1e6, wr / 1e6, len(v["FETCH_SIZE"])))
print("algorithmic per direction, %d projections of 2560 x 2560: 2-byte %.1f MB, 4-byte %.1f MB, 8-byte %.1f MB" % (D, D * 2560 * 2560 * 2 / 1e6, D * 2560 * 2560 * 4 / 1e6, D * 2560 * 2560 * 8 / 1e6))
PY
