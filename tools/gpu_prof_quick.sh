#!/bin/bash
# GPU box: rocprofv3 kernel statistics of a short bench run of one workload -> gpurun_out/TAG_WL_kernel_stats.csv (top rows printed)
tag=${1:-r06_x}; wl=${2:-config2}; steps=${3:-100}
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_q
rocprofv3 --kernel-trace --stats -d /tmp/prof_q -o p -- python $root/bench.py --workload $wl --steps $steps --no-extras --no-cpu-baseline > /tmp/prof_q.log 2>&1
python $root/tools/rocpd_stats.py $(find /tmp/prof_q -name '*.db' | head -1) $root/gpurun_out/${tag}_${wl}_kernel_stats.csv
python - <<P
import csv
rows=[r for r in csv.reader(open('$root/gpurun_out/${tag}_${wl}_kernel_stats.csv')) if not r[0].startswith('#')]
for r in rows[:16]:
    print('%-70s %7s %10s %8s %6s' % (r[0][:70], r[1], r[2], r[3], r[6]))
P
