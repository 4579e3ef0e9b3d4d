#!/bin/bash
# rocprofv3 kernel statistics of the closed-loop filter session (run on the GPU box, from the repo root)
# usage: tools/prof_session.sh OUT.csv [session_timing args]
out=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_sess
OVP_TIMING_MODES=0 rocprofv3 --kernel-trace --stats -d /tmp/prof_sess -o sess -- python $GRAFT_REPO_ROOT/tools/session_timing.py "$@" > /tmp/prof_sess.log 2>&1
db=$(find /tmp/prof_sess -name '*.db' | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $db $GRAFT_REPO_ROOT/$out
