#!/bin/bash
# On the GPU box (from the repo root): kernel statistics of K1 with B in LDS (default) and through the scratch (OVP_K1_BSCR=1), same
# commands, under rocprofv3 --kernel-trace --stats.  usage: tools/gpu_k1_prof.sh TAG [workloads...]   (writes gpurun_out/TAG_*)
tag=${1:-k1}; shift
wls=${@:-config3 config2}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for wl in $wls; do
  steps=50; [ $wl = config4 ] && steps=10
  for variant in lds scratch; do
    if [ $variant = scratch ]; then export OVP_K1_BSCR=1; else unset OVP_K1_BSCR; fi
    name=${wl}_${variant}
    rm -rf /tmp/prof_$name
    timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o p -- python $root/bench.py --workload $wl --steps $steps --no-extras --no-cpu-baseline > $out/${tag}_${name}_bench.json 2> /tmp/prof_$name.log
    db=$(find /tmp/prof_$name -name '*.db' | head -1)
    python $root/tools/rocpd_stats.py $db $out/${tag}_${name}_kernel_stats.csv > /dev/null 2>&1
    echo "== $name"; grep -E "k_feat|k_gram_pair" $out/${tag}_${name}_kernel_stats.csv | cut -d, -f1-6 | cut -c1-150
    python -c "import json,sys; d=json.loads(open('$out/${tag}_${name}_bench.json').read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])" 2>/dev/null
  done
done
unset OVP_K1_BSCR
