mkdir -p gpurun_out/r05f
bash tools/profile.sh r05f > gpurun_out/r05f/profile.log 2>&1
# keep what is judged (summary, kernel stats of the timing pass, digest); the raw per-launch CSVs of twelve passes exceed what gpurun copies back
find gpurun_out/prof_r05f -name "*.csv" ! -name "out_kernel_stats.csv" -delete
find gpurun_out/prof_r05f -name "*.db" -delete
du -sh gpurun_out/prof_r05f | tail -1
python bench.py > gpurun_out/r05f/bench_1gpu.json 2> gpurun_out/r05f/bench_1gpu.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05f/bench_driver_cmd.json 2> gpurun_out/r05f/bench_driver_cmd.err
timeout 1500 python tools/fuzz_parity.py 40000 5101 2>&1 | tail -2 > gpurun_out/r05f/fuzz_parity.txt
DCP_LIB_PATH=discorpy_amd/lib/libdiscorpy_hip_bounds.so timeout 900 python tools/fuzz_parity.py 12000 5102 --bounds 2>&1 | tail -3 > gpurun_out/r05f/fuzz_bounds.txt
FUZZ_BIG=1 timeout 900 python tools/fuzz_parity.py 1500 5103 2>&1 | tail -2 > gpurun_out/r05f/fuzz_big.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05f/smoke.txt 2>&1
for f in fuzz_parity fuzz_bounds fuzz_big smoke; do tail -n 3 gpurun_out/r05f/$f.txt | cut -c1-700; done
head -c 1500 gpurun_out/r05f/bench_driver_cmd.json; echo
du -sh gpurun_out
