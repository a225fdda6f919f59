#!/bin/bash
# What kind of box is this?  Sustained cfg2 kernel time (work-group box and per-wave box), nearest, with clocks and power sampled under load.
rocm-smi --showmaxpower --showperflevel --showmemvendor 2>/dev/null | grep -E "GPU\[0\]" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ';'; echo
cat > /tmp/opt_wg0.py <<'PY'
PY
for spec in "wg 1 1 1" "perwave 0 1 1" "wg_nearest 1 0 0" "wg_f32 1 2 1"; do
  set -- $spec
  DCP_WG=$2 python - $3 $4 > /tmp/spin_$1.log 2>&1 <<'PY' &
import os, sys, time
sys.path.insert(0, os.getcwd())
from discorpy_amd import _ffi as F
F.lib(); F.set_option("wg_box", int(os.environ["DCP_WG"]))
sys.argv = ["spin", "6", sys.argv[1], sys.argv[2]]
exec(open("tools/spin_k1.py").read())
PY
  P=$!
  sleep 3.5
  S=""
  for i in 1 2 3; do S="$S $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|fclk|Power \(W\)' | sed -E 's/.*(sclk|fclk) clock level: [0-9]+: \(([0-9]+)Mhz\).*/\1=\2/; s/.*Power \(W\): ([0-9.]+).*/\1W/' | tr '\n' ' ')"; sleep 0.6; done
  wait $P
  echo "$1: $(grep 'us per' /tmp/spin_$1.log) | $S"
done
