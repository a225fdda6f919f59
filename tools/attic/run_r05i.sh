mkdir -p gpurun_out/r05i
PASSES="trace sq1 fetch write sfetch swrite" bash tools/profile.sh r05i > gpurun_out/r05i/profile.log 2>&1
find gpurun_out/prof_r05i -name "*.csv" ! -name "out_kernel_stats.csv" -delete
find gpurun_out/prof_r05i -name "*.db" -delete
python bench.py > gpurun_out/r05i/bench_1gpu.json 2> gpurun_out/r05i/bench_1gpu.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05i/bench_driver_cmd.json 2> gpurun_out/r05i/bench_driver_cmd.err
timeout 900 python tools/fuzz_parity.py 20000 5301 2>&1 | tail -2 > gpurun_out/r05i/fuzz_parity.txt
tail -n 2 gpurun_out/r05i/fuzz_parity.txt | cut -c1-300
head -c 700 gpurun_out/r05i/bench_driver_cmd.json; echo
head -12 gpurun_out/prof_r05i/summary.txt | cut -c1-400
du -sh gpurun_out
