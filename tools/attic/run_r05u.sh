mkdir -p gpurun_out/r05u
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05u/pytest_gpu.txt 2>&1; grep -n "passed\|failed" gpurun_out/r05u/pytest_gpu.txt | tail -2
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05u/smoke.txt 2>&1; tail -n 1 gpurun_out/r05u/smoke.txt | cut -c1-300
PASSES="trace sq1 fetch write sfetch swrite" bash tools/profile.sh r05u > gpurun_out/r05u/profile.log 2>&1
find gpurun_out/prof_r05u -name "*.csv" ! -name "out_kernel_stats.csv" -delete
find gpurun_out/prof_r05u -name "*.db" -delete
python bench.py > gpurun_out/r05u/bench_1gpu.json 2> gpurun_out/r05u/bench_1gpu.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05u/bench_driver_cmd.json 2> gpurun_out/r05u/bench_driver_cmd.err
head -c 400 gpurun_out/r05u/bench_driver_cmd.json; echo
head -12 gpurun_out/prof_r05u/summary.txt | cut -c1-200
for s in 51 52; do timeout 700 python tools/fuzz_pf2d.py 1500 $s 2>&1 | tail -1; done > gpurun_out/r05u/fuzz_pf2d.txt
FUZZ_BIG=1 timeout 900 python tools/fuzz_parity.py 4000 5901 2>&1 | tail -1 > gpurun_out/r05u/fuzz_big.txt
timeout 900 python tools/fuzz_parity.py 30000 5902 2>&1 | tail -1 > gpurun_out/r05u/fuzz_parity.txt
cut -c1-250 gpurun_out/r05u/fuzz_*.txt
