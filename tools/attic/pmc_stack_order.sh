#!/bin/bash
# L2 fetch bytes of stack_wg_kernel under the two tile orders (tools/ab_stack_order.py, one mode per rocprofv3 counter pass).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp AB_REPS=1 AB_LAUNCHES=3 AB_SETTLE_MS=1
cd /tmp
for mode in 2 0; do
  for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $c | cut -d' ' -f1)
    rm -rf /tmp/pso_${mode}_$tag
    AB_MODES=$mode timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pso_${mode}_$tag -o out -- python $ROOT/tools/ab_stack_order.py ${1:-256} > /tmp/pso_${mode}_$tag.log 2>&1
    python - $mode /tmp/pso_${mode}_$tag <<'PY'
import csv, glob, sys, collections
mode, d = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/out_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "stack_wg_kernel" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print("xcd_remap=%s  %-42s %-14s mean per launch %.4g (n=%d)" % (mode, k, c, sum(v) / len(v), len(v)))
PY
  done
done
