#!/bin/bash
# Builds experiment variants of the library: tools/variants.sh name "-DFLAG=.." [name2 "-D.."] ...
set -e
cd "$(dirname "$0")/../discorpy_amd/csrc"
while [ $# -ge 2 ]; do
  n=$1; f=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $f -c unwarp_kernels.hip -o /tmp/uk_$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libdcp_var_$n.so /tmp/uk_$n.o ../lib/spline_kernels.o ../lib/typed_kernels.o ../lib/api_core.o ../lib/api_image.o ../lib/api_stack.o ../lib/api_spline.o ../lib/api_rccl.o -pthread -ldl
done
