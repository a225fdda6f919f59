#!/bin/bash
TL=$(python -c "import torch,os;print(os.path.dirname(torch.__file__)+'/lib')" 2>/dev/null)
for e in "X=1" "HSA_REV_COPY_DIR=1" "HSA_ENABLE_SDMA_RECOMMENDED_ENG=0" "HSA_ENABLE_SDMA_RECOMMENDED_ENG=1" "HSA_ENABLE_SDMA_GANG=0" "HSA_ENABLE_SDMA_GANG=1" "HSA_ENABLE_SDMA=0" "GPU_MAX_HW_QUEUES=8" "DEBUG_HIP_DYNAMIC_QUEUES=0"; do
  echo "== torch-bundled runtime, $e"
  env $e LD_PRELOAD=$TL/libamdhip64.so:$TL/libhsa-runtime64.so tools/attic/ubench_host/two_copy 2>&1 | grep -v amdgpu.ids | sed -n 1,3p
done
echo "== /opt/rocm runtime"
tools/attic/ubench_host/two_copy 2>&1 | grep -v amdgpu.ids | sed -n 1,3p
