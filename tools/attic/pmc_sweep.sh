#!/bin/bash
# tools/pmc_sweep.sh <tag> <blend> <order> [lib]: SQ / LDS / TA / TCP counter passes (each its own rocprofv3 run, --kernel-trace
# only) over a short run of the cfg2 frame kernel; prints per-launch averages of the last half of the dispatches.
TAG=$1; BL=$2; ORD=$3; LIB=${4:-}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
[ -n "$LIB" ] && export DCP_LIB_PATH=$LIB
cd /tmp
CMD="python $ROOT/tools/spin_k1.py 0.0 $BL $ORD"
run() { local name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o out -- $CMD > "$OUT/$name.log" 2>&1; echo "pass $name rc=$?"; }
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY
run sq2 SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE
run sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL
run sq4 SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM
run tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
run tcp2 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
run tcp3 TCP_TA_TCP_STATE_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_UTCL1_REQUEST_sum
cd "$ROOT"
python - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
dur = defaultdict(list)
for f in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        dur[row["Kernel_Name"]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
for k, cs in acc.items():
    if "remap" not in k:
        continue
    d = sorted(dur.get(k, [0]))
    print("KERNEL", k[:90], "median_ns(profiled)", d[len(d) // 2], "n", len(d))
    for c in sorted(cs):
        v = cs[c][len(cs[c]) // 2:]
        print("    %-40s %16.1f" % (c, sum(v) / len(v)))
PY
