#!/bin/bash
# The profile set of a round, on the GPU box (from the repo root): usage tools/prof_round.sh TAG   (writes gpurun_out/TAG_*)
# kernel statistics under rocprofv3 (config 3 = the headline, config 2, config 4), HBM traffic and SQ counters as separate --pmc
# passes (never combined with a trace domain), the bench lines themselves, the closed-loop session.
tag=${1:-r06_final}
root=$GRAFT_REPO_ROOT
[ -z "$root" ] && root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
stats() {  # name, command...
  name=$1; shift
  rm -rf /tmp/prof_$name
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o p -- "$@" > /tmp/prof_$name.log 2>&1
  db=$(find /tmp/prof_$name -name '*.db' | head -1)
  python $root/tools/rocpd_stats.py $db $out/${tag}_${name}_kernel_stats.csv
}
# 1. the driver's command, untouched
python $root/bench.py > $out/${tag}_bench_driver_cmd.json 2> $out/${tag}_bench_driver_cmd.err
# 2. kernel statistics
rm -rf /tmp/prof_c3
rocprofv3 --kernel-trace --stats -d /tmp/prof_c3 -o p -- python $root/bench.py --steps 100 --no-cpu-baseline > $out/${tag}_bench_under_rocprof.json 2> /tmp/prof_c3.log
python $root/tools/rocpd_stats.py $(find /tmp/prof_c3 -name '*.db' | head -1) $out/${tag}_kernel_stats.csv
stats config2 python $root/bench.py --workload config2 --steps 100 --no-extras --no-cpu-baseline
stats config4 python $root/bench.py --workload config4 --steps 20 --no-extras --no-cpu-baseline
# 3. counters: separate passes, kernel-trace only
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$ctr
  rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_$ctr -o p --output-format csv -- python $root/bench.py --steps 20 --no-extras --no-cpu-baseline > /tmp/pmc_$ctr.log 2>&1
done
python $root/tools/pmc_summary.py $out/${tag}_hbm_traffic_pmc.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over python bench.py --steps 20 --no-extras --no-cpu-baseline; KB per launch as reported (the guide's gfx950 correction doubles FETCH_SIZE for wide coalesced reads)" /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE > /dev/null
rm -rf /tmp/pmc_sq
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d /tmp/pmc_sq -o p --output-format csv -- python $root/bench.py --steps 20 --no-extras --no-cpu-baseline > /tmp/pmc_sq.log 2>&1
python $root/tools/pmc_summary.py $out/${tag}_sq_counters_pmc.json "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES over python bench.py --steps 20 --no-extras --no-cpu-baseline; per launch" /tmp/pmc_sq > /dev/null
# 4. closed loop
cd $root
OVP_TIMING_MODES=0 python tools/session_timing.py --out $out/${tag}_session_timing.json > /dev/null 2>&1
OVP_TIMING_MODES=0 python tools/session_timing.py --planes 2 --out $out/${tag}_session_planes_timing.json > /dev/null 2>&1
tools/prof_session.sh gpurun_out/${tag}_session_kernel_stats.csv
ls -la $out | grep $tag
