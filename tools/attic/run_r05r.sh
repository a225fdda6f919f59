mkdir -p gpurun_out/r05r
for seed in 5801 5802 5803 5804 5805; do timeout 1200 python tools/fuzz_parity.py 40000 $seed 2>&1 | tail -1; done > gpurun_out/r05r/fuzz_parity.txt
FUZZ_BIG=1 timeout 1200 python tools/fuzz_parity.py 6000 5806 2>&1 | tail -1 > gpurun_out/r05r/fuzz_big.txt
cut -c1-220 gpurun_out/r05r/*.txt
