for rep in 1 2; do
for lib in discorpy_amd/lib/libdiscorpy_hip.so discorpy_amd/lib/v8/libdiscorpy_hip.so; do
  echo "== $lib"; DCP_LIB_PATH=$lib timeout 200 python tools/check_pf2d.py --skip-parity --chunks 0 2>&1 | grep "fused prefilter" | head -4
done; done
timeout 300 python tools/check_pf2d.py --skip-timing 2>&1 | grep -c " OK"
timeout 300 python tools/check_pf2d.py --skip-timing 2>&1 | grep "BAD" | grep -v "causal / anticausal" | head -3
