#!/bin/bash
# A/B of library variants (discorpy_amd/lib/variants/lib_*.so, tools/variants.sh) under the two-stream dispatch: every variant twice, alternated
for rep in 1 2; do
  for lib in discorpy_amd/lib/variants/lib_*.so; do
    n=$(basename $lib .so)
    DCP_LIB_PATH=$PWD/$lib python tools/time_dispatch.py --rounds 1 --modes ordered,two_streams 2>&1 | grep "round" | sed "s/^/$n  /"
  done
done
