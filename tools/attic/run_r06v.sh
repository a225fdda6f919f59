mkdir -p gpurun_out/r06v
timeout 1500 python tools/fuzz_parity.py 100000 7001 2>&1 | tail -1 > gpurun_out/r06v/fuzz_parity.txt
FUZZ_BIG=1 timeout 900 python tools/fuzz_parity.py 6000 7002 2>&1 | tail -1 > gpurun_out/r06v/fuzz_big.txt
timeout 900 python tools/fuzz_parity.py 12000 7003 --bounds 2>&1 | tail -1 > gpurun_out/r06v/fuzz_bounds_build.txt
for sd in 71 72; do timeout 700 python tools/fuzz_pf2d.py 1500 $sd 2>&1 | tail -1; done > gpurun_out/r06v/fuzz_pf2d.txt
cut -c1-300 gpurun_out/r06v/*.txt
