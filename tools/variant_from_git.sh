#!/bin/bash
# A/B reference: libdcp_var_<name>.so with unwarp_kernels.hip taken from a git revision (default HEAD) and everything else from the
# working tree (same C ABI, so the current Python binding loads it):  tools/variant_from_git.sh name [rev] ["-DFLAGS"]
case "${1:-}" in -h|--help) sed -n '2,3p' "$0" | sed 's/^# \{0,1\}//'; exit 0;; esac
set -e
N=$1; REV=${2:-HEAD}; FLAGS=${3:-}
cd "$(dirname "$0")/../discorpy_amd/csrc"
git show $REV:discorpy_amd/csrc/unwarp_kernels.hip > /tmp/uk_$N.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I. $FLAGS -c /tmp/uk_$N.hip -o /tmp/uk_$N.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libdcp_var_$N.so /tmp/uk_$N.o ../lib/spline_kernels.o ../lib/typed_kernels.o ../lib/api_core.o ../lib/api_image.o ../lib/api_stack.o ../lib/api_spline.o ../lib/api_rccl.o -pthread -ldl
