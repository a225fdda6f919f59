mkdir -p gpurun_out/r05b
for rep in 1 2 3; do
for c in 1 4 19 60; do
python bench.py --gpus 1 --steps 20 --warmup 5 --cycles $c --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('cycles',$c,'launch_us',j['launch_us'],'wall_ms',j['timed_region_ms'],'value',j['value'],'value_wall',j['value_wall'],'enq',j['host_enqueue_us_per_launch'],'sync',j['sync_ms'],'med',j['launch_us_median'])
"
done
done 2>&1 | tee gpurun_out/r05b/ab_cycles.txt
