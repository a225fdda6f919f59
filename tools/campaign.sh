#!/bin/bash
# One GPU-box campaign, parameterised (replaces the eleven run_r05*.sh of round 5: tools/attic/):
#   gpurun --timeout 3000 -- 'bash tools/campaign.sh <tag> [step ...]'
# writes everything under gpurun_out/<tag>/ (scratch; tools/save_profile.py copies the judged part into profiles/).  Steps, default all:
#   tests         pytest -m gpu (the round-end suite)              smoke        __graft_entry__.smoke()
#   profile       tools/profile.sh <tag>: rocprofv3 --kernel-trace --stats of the default bench command + the PMC passes
#   profile_ordered  the same timing pass with --dispatch ordered (one stream: the kernel's average duration by itself)
#   bench         python bench.py                                  bench_driver  the driver's command, --gpus 1 --steps 20 --warmup 5
#   bench_ordered the driver's command with --dispatch ordered     dispatch     tools/time_dispatch.py (ordered / any-order / N streams / batch)
#   spline        tools/time_spline_modes.py + per-kernel times    pmc_spline   tools/pmc_spline.sh
#   fuzz          30 000 randomised cases against the oracle       fuzz_big     4 000 large frames       fuzz_pf2d  3 000 aimed at the spline prefilter
case "${1:-}" in -h|--help) sed -n '2,11p' "$0" | sed 's/^# \{0,1\}//'; exit 0;; esac
set -u
TAG=${1:-run}; shift || true
STEPS=${*:-tests smoke profile profile_ordered bench bench_driver bench_ordered dispatch spline fuzz_pf2d fuzz_big fuzz}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
O=gpurun_out/$TAG
mkdir -p $O
for s in $STEPS; do
  echo "=== $s"
  case $s in
    tests) timeout 1800 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_gpu.txt 2>&1; grep -n "passed\|failed" $O/pytest_gpu.txt | tail -2 ;;
    smoke) python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -n 1 $O/smoke.txt | cut -c1-300 ;;
    profile) PASSES="${PASSES:-trace sq1 fetch write sfetch swrite}" bash tools/profile.sh $TAG > $O/profile.log 2>&1
             find gpurun_out/prof_$TAG -name "*.csv" ! -name "out_kernel_stats.csv" -delete; find gpurun_out/prof_$TAG -name "*.db" -delete
             head -12 gpurun_out/prof_$TAG/summary.txt | cut -c1-200 ;;
    profile_ordered) PASSES="trace" bash tools/profile.sh ${TAG}_ordered --dispatch ordered > $O/profile_ordered.log 2>&1
             find gpurun_out/prof_${TAG}_ordered -name "*.csv" ! -name "out_kernel_stats.csv" -delete; find gpurun_out/prof_${TAG}_ordered -name "*.db" -delete
             head -8 gpurun_out/prof_${TAG}_ordered/summary.txt | cut -c1-200 ;;
    bench) python bench.py > $O/bench_1gpu.json 2> $O/bench_1gpu.err; head -c 300 $O/bench_1gpu.json; echo ;;
    bench_driver) python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; head -c 400 $O/bench_driver_cmd.json; echo ;;
    bench_ordered) python bench.py --gpus 1 --steps 20 --warmup 5 --dispatch ordered --no-extras --no-cpu-baseline > $O/bench_ordered.json 2> $O/bench_ordered.err; head -c 300 $O/bench_ordered.json; echo ;;
    dispatch) python tools/time_dispatch.py --rounds 3 --modes ordered,two_streams,streams3,any_order,batch > $O/dispatch_modes.txt 2>&1; tail -15 $O/dispatch_modes.txt ;;
    spline) python tools/time_spline_modes.py > $O/time_spline_modes.txt 2>&1; tail -30 $O/time_spline_modes.txt ;;
    pmc_spline) bash tools/pmc_spline.sh > $O/pmc_spline.txt 2>&1; tail -12 $O/pmc_spline.txt ;;
    fuzz) timeout 900 python tools/fuzz_parity.py 30000 ${SEED:-6902} 2>&1 | tail -1 > $O/fuzz_parity.txt; cut -c1-250 $O/fuzz_parity.txt ;;
    fuzz_big) FUZZ_BIG=1 timeout 900 python tools/fuzz_parity.py 4000 ${SEED:-6901} 2>&1 | tail -1 > $O/fuzz_big.txt; cut -c1-250 $O/fuzz_big.txt ;;
    fuzz_pf2d) for sd in 61 62; do timeout 700 python tools/fuzz_pf2d.py 1500 $sd 2>&1 | tail -1; done > $O/fuzz_pf2d.txt; cut -c1-250 $O/fuzz_pf2d.txt ;;
    *) echo "unknown step $s" ;;
  esac
done
