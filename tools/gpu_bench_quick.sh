#!/bin/bash
# GPU box: the config-3 step (50 steps, no CPU leg, no side figures) and the plane-loop tests - a quick A/B line
wl=${1:-config3}
timeout 120 python bench.py --workload $wl --steps ${2:-50} --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
dc=d.get('device_clock',{})
print('$wl ms_per_step %.4f  plane_loop %.4f  point_update %.4f' % (d['ms_per_step'], dc.get('plane_loop_ms',0), dc.get('point_update_ms',0)))
r=d.get('roofline',{})
print('   dominant kernel avg launch ms', r.get('avg_launch_ms'))"
