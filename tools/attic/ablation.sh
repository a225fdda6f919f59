mkdir -p gpurun_out/r02a
L=$PWD/discorpy_amd/lib
( for v in "" var_nowait var_nofill var_nostore var_neither ""; do
  if [ -z "$v" ]; then python tools/time_k1.py full; else DCP_LIB_PATH=$L/libdcp_$v.so python tools/time_k1.py $v; fi
done ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r02a/ablation.log
# clocks and power during a sustained run of the full kernel
python bench.py --steps 12000 --warmup 5 --no-cpu-baseline > /tmp/b.log 2>&1 &
BP=$!
sleep 6
for i in 1 2 3 4 5; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power \(W\)" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ' '; echo; sleep 1; done > gpurun_out/r02a/clocks.log
wait $BP
tail -1 /tmp/b.log > gpurun_out/r02a/bench_long.json
cat gpurun_out/r02a/ablation.log gpurun_out/r02a/clocks.log
