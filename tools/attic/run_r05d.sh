mkdir -p gpurun_out/r05d
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "cfg3 or fused" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_debug_builds.py tests/test_losa_hdf.py tests/test_rccl_abi.py tests/test_rccl_world.py -m gpu -q 2>&1 | tail -5
python tools/time_cfg3.py 2>&1 | tee gpurun_out/r05d/time_cfg3.txt
FUZZ_STAGED_KIND=fused timeout 900 python tools/fuzz_parity.py 1500 501 2>&1 | tail -3 | tee gpurun_out/r05d/fuzz_fused.txt
FUZZ_STAGED_KIND=fused DCP_LIB_PATH=discorpy_amd/lib/libdiscorpy_hip_bounds.so timeout 900 python tools/fuzz_parity.py 600 502 --bounds 2>&1 | tail -3 | tee gpurun_out/r05d/fuzz_fused_bounds.txt
bash tools/pmc_spline.sh 2>&1 | tail -12 | tee gpurun_out/r05d/pmc_spline.txt
