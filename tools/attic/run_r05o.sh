mkdir -p gpurun_out/r05o
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05o/pytest_gpu.txt 2>&1; tail -n 3 gpurun_out/r05o/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05o/smoke.txt 2>&1; tail -n 1 gpurun_out/r05o/smoke.txt | cut -c1-300
PASSES="trace sq1 fetch write sfetch swrite" bash tools/profile.sh r05o > gpurun_out/r05o/profile.log 2>&1
find gpurun_out/prof_r05o -name "*.csv" ! -name "out_kernel_stats.csv" -delete
find gpurun_out/prof_r05o -name "*.db" -delete
python bench.py > gpurun_out/r05o/bench_1gpu.json 2> gpurun_out/r05o/bench_1gpu.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05o/bench_driver_cmd.json 2> gpurun_out/r05o/bench_driver_cmd.err
timeout 900 python tools/fuzz_parity.py 20000 5701 2>&1 | tail -1 > gpurun_out/r05o/fuzz_parity.txt
cut -c1-300 gpurun_out/r05o/fuzz_*.txt
head -c 600 gpurun_out/r05o/bench_driver_cmd.json; echo
head -12 gpurun_out/prof_r05o/summary.txt | cut -c1-300
