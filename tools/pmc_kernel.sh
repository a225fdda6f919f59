#!/bin/bash
# SQ counters of every kernel a command launches whose name contains $KERNELS (comma separated), averaged per launch:
#   KERNELS=spline_prefilter2d,spline_wg_kernel tools/pmc_kernel.sh python tools/time_spline.py --orders 3
case "${1:-}" in -h|--help) sed -n '2,3p' "$0" | sed 's/^# \{0,1\}//'; exit 0;; esac
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_INSTS_SALU" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pk_$i
  (cd $ROOT && timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pk_$i -o out -- "$@" > /tmp/pk_$i.log 2>&1)
done
python - <<'PY'
import csv, glob, os, collections
want = [w for w in os.environ.get("KERNELS", "").split(",") if w]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pk_*/**/out_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void dcp::", "")
        if not want or any(w in k for w in want):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(k[:90])
    for c, vals in sorted(v.items()):
        print("    %-28s %16.1f  (n=%d)" % (c, sum(vals) / len(vals), len(vals)))
PY
