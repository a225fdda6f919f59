#!/bin/bash
# Builds experiment variants of the library into discorpy_amd/lib/variants/lib_<name>.so (what tools/ab_variants.sh alternates):
#   tools/variants.sh name "-DFLAG=.." [name2 "-D.."] ...
# FILE=spline_kernels.hip tools/variants.sh ...  rebuilds that translation unit instead of unwarp_kernels.hip.
case "${1:-}" in -h|--help) sed -n '2,3p' "$0" | sed 's/^# \{0,1\}//'; exit 0;; esac
set -e
cd "$(dirname "$0")/../discorpy_amd/csrc"
FILE=${FILE:-unwarp_kernels.hip}
BASE=${FILE%.hip}
while [ $# -ge 2 ]; do
  n=$1; f=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden $f -c $FILE -o /tmp/var_$n.o
  OBJS=""
  for o in unwarp_kernels spline_kernels typed_kernels color_kernels api_core api_image api_stack api_spline api_rccl; do
    if [ "$o" = "$BASE" ]; then OBJS="$OBJS /tmp/var_$n.o"; else OBJS="$OBJS ../lib/$o.o"; fi
  done
  mkdir -p ../lib/variants
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -Wl,--version-script=exports.map -o ../lib/variants/lib_$n.so $OBJS -ldl
done
