#!/bin/bash
# rocprofv3 passes over bench.py (run on the GPU box through gpurun).  Usage:
#   tools/profile.sh <tag> [extra bench.py args]
# Writes gpurun_out/prof_<tag>/<pass>/..., then tools/summarize_prof.py condenses them.
# Timing and counters are collected in SEPARATE runs (PMC passes carry --kernel-trace only).
case "${1:-}" in -h|--help) sed -n '2,5p' "$0" | sed 's/^# \{0,1\}//'; exit 0;; esac
set -u
TAG=${1:-run}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
# what the counters belong to: the digest of the kernel sources of THIS tree (bench.py compares it with the tree it times)
(cd "$ROOT" && python -c "import bench; print(bench.kernel_sources_digest())" > "$OUT/kernel_sources.sha256" 2>/dev/null)
export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras $*"      # counter passes: the headline kernel only
BENCH_FULL="python $ROOT/bench.py --no-cpu-baseline $*"    # the timing pass profiles the default command: headline + every other configuration
cd /tmp
run() { # name, rocprof args...   (PASSES="trace fetch" restricts the passes that run)
  local name=$1; shift
  if [ -n "${PASSES:-}" ] && ! echo " $PASSES " | grep -q " $name "; then return; fi
  local cmd=$BENCH
  if [ "$name" = trace ]; then cmd=$BENCH_FULL; fi
  timeout 400 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -o out -- $cmd > "$OUT/$name.log" 2>&1
  echo "pass $name rc=$?"
}
run trace --kernel-trace --stats
run sq1 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES
run sq2 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_LDS GRBM_GUI_ACTIVE
run fetch --kernel-trace --pmc FETCH_SIZE
run write --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
# HBM traffic of the stack kernel too: config 4, one 8-GPU shard (256 projections, every row)
BENCH_SAVE=$BENCH
BENCH="python $ROOT/bench.py --workload stack --depth 256 --steps 2 --warmup 1"
run sfetch --kernel-trace --pmc FETCH_SIZE
run swrite --kernel-trace --pmc WRITE_SIZE
BENCH=$BENCH_SAVE
run tcc --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
run tcc2 --kernel-trace --pmc TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_EA0_RDREQ_DRAM_sum
run tcp --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
run tcp2 --kernel-trace --pmc TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
# (TA_* / TD_* counter passes hang rocprofv3 on this pool -- not collected)
run sq3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT
# calibration of FETCH_SIZE / WRITE_SIZE on kernels that move exactly 1 GiB each way
if [ -n "${CALIB:-1}" ]; then
  hipcc --offload-arch=gfx950 -O3 -o /tmp/calib_copy $ROOT/tools/calib_copy.hip > "$OUT/calib_build.log" 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/calib_$c" -o out -- /tmp/calib_copy > "$OUT/calib_$c.log" 2>&1
    echo "pass calib_$c rc=$?"
  done
fi
cd "$ROOT"
python tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
