#!/bin/bash
# Core clock and package power while the bench kernel runs back to back (rocm-smi samples during a ~15 s run).
python bench.py --steps 18000 --warmup 5 --no-cpu-baseline > /tmp/b.log 2>&1 &
BP=$!
sleep 5
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ' '; echo; sleep 1.2; done
wait $BP
python - <<'PY'
import json
b = json.loads(open("/tmp/b.log").read().strip().split("\n")[-1])
print("bench over that run: %.2f us per launch, %.0f Mpixels/s" % (b["roofline"]["launch_us"], b["value"]))
PY
