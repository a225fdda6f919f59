mkdir -p gpurun_out/r05m
for seed in 5601 5602 5603; do timeout 1200 python tools/fuzz_parity.py 40000 $seed 2>&1 | tail -1; done > gpurun_out/r05m/fuzz_parity.txt
for kind in fused stack color batch radial persp; do FUZZ_STAGED_KIND=$kind DCP_LIB_PATH=discorpy_amd/lib/libdiscorpy_hip_bounds.so timeout 600 python tools/fuzz_parity.py 4000 56$RANDOM --bounds 2>&1 | tail -2; done > gpurun_out/r05m/fuzz_staged_bounds.txt
FUZZ_BIG=1 FUZZ_ONLY=spline timeout 900 python tools/fuzz_parity.py 6000 5605 2>&1 | tail -1 > gpurun_out/r05m/fuzz_spline_big.txt
cut -c1-260 gpurun_out/r05m/*.txt
