#!/bin/bash
# Builds ablated copies of the library (experiments only) next to the real one:
#   lib/libdcp_ablate{1,2,3}.so  = no gather loads / no store / neither
set -e
cd "$(dirname "$0")/../discorpy_amd/csrc"
for a in 1 2 3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DDCP_ABLATE=$a -c unwarp_kernels.hip -o /tmp/uk_ab$a.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libdcp_ablate$a.so /tmp/uk_ab$a.o ../lib/unwarp_api.o
done
