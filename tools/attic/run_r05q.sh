mkdir -p gpurun_out/r05q
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05q/pytest_gpu.txt 2>&1; grep -n "passed\|failed" gpurun_out/r05q/pytest_gpu.txt | tail -2
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05q/smoke.txt 2>&1; tail -n 1 gpurun_out/r05q/smoke.txt | cut -c1-300
PASSES="trace sq1 fetch write sfetch swrite" bash tools/profile.sh r05q > gpurun_out/r05q/profile.log 2>&1
find gpurun_out/prof_r05q -name "*.csv" ! -name "out_kernel_stats.csv" -delete
find gpurun_out/prof_r05q -name "*.db" -delete
python bench.py > gpurun_out/r05q/bench_1gpu.json 2> gpurun_out/r05q/bench_1gpu.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05q/bench_driver_cmd.json 2> gpurun_out/r05q/bench_driver_cmd.err
head -c 400 gpurun_out/r05q/bench_driver_cmd.json; echo
head -12 gpurun_out/prof_r05q/summary.txt | cut -c1-200
