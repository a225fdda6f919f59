mkdir -p gpurun_out/r05l
timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/r05l/pytest_gpu.txt 2>&1; tail -n 16 gpurun_out/r05l/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05l/smoke.txt 2>&1; tail -n 2 gpurun_out/r05l/smoke.txt | cut -c1-400
PASSES="trace sq1 fetch write sfetch swrite" bash tools/profile.sh r05l > gpurun_out/r05l/profile.log 2>&1
find gpurun_out/prof_r05l -name "*.csv" ! -name "out_kernel_stats.csv" -delete
find gpurun_out/prof_r05l -name "*.db" -delete
tools/pmc_spline.sh > gpurun_out/r05l/pmc_spline.txt 2>&1; cp gpurun_out/pmc_spline.json gpurun_out/r05l/pmc_spline.json
python bench.py > gpurun_out/r05l/bench_1gpu.json 2> gpurun_out/r05l/bench_1gpu.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05l/bench_driver_cmd.json 2> gpurun_out/r05l/bench_driver_cmd.err
timeout 900 python tools/fuzz_parity.py 30000 5501 2>&1 | tail -1 > gpurun_out/r05l/fuzz_parity.txt
FUZZ_ONLY=spline,color,coords timeout 600 python tools/fuzz_parity.py 10000 5502 2>&1 | tail -1 > gpurun_out/r05l/fuzz_spline_color_coords.txt
FUZZ_BIG=1 timeout 900 python tools/fuzz_parity.py 3000 5503 2>&1 | tail -1 > gpurun_out/r05l/fuzz_big.txt
cut -c1-300 gpurun_out/r05l/fuzz_*.txt
head -c 900 gpurun_out/r05l/bench_driver_cmd.json; echo
head -12 gpurun_out/prof_r05l/summary.txt | cut -c1-300
tail -3 gpurun_out/r05l/pmc_spline.txt
du -sh gpurun_out
