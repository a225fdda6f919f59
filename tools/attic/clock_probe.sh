#!/bin/bash
# tools/clock_probe.sh <label> <lib or ""> <blend> <order>: core clock / power while the cfg2 kernel of that build runs back to back
LBL=$1; LIB=$2; BL=$3; ORD=$4
if [ -n "$LIB" ]; then export DCP_LIB_PATH=$LIB; fi
python tools/spin_k1.py 7 $BL $ORD > /tmp/spin.log 2>&1 &
P=$!
sleep 4
S=""
for i in 1 2 3; do S="$S $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Power \(W\)' | sed -E 's/.*\(([0-9]+)Mhz\).*/\1MHz/; s/.*Power \(W\): ([0-9.]+).*/\1W/' | tr '\n' ' ')"; sleep 0.7; done
wait $P
echo "$LBL: $(grep 'us per' /tmp/spin.log) | $S"
