mkdir -p gpurun_out/r05j
for seed in 5401 5402 5403 5404; do timeout 1500 python tools/fuzz_parity.py 40000 $seed 2>&1 | tail -1; done > gpurun_out/r05j/fuzz_parity.txt
for kind in fused stack color batch radial persp; do FUZZ_STAGED_KIND=$kind DCP_LIB_PATH=discorpy_amd/lib/libdiscorpy_hip_bounds.so timeout 900 python tools/fuzz_parity.py 6000 55$RANDOM --bounds 2>&1 | tail -2; done > gpurun_out/r05j/fuzz_staged_bounds.txt
FUZZ_BIG=1 timeout 1500 python tools/fuzz_parity.py 4000 5405 2>&1 | tail -1 > gpurun_out/r05j/fuzz_big.txt
FUZZ_ONLY=spline,color,coords timeout 900 python tools/fuzz_parity.py 12000 5406 2>&1 | tail -1 > gpurun_out/r05j/fuzz_spline_color_coords.txt
cut -c1-260 gpurun_out/r05j/*.txt
