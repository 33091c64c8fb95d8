#!/bin/bash
# rocprofv3 kernel-trace summary of scripts/time_configs.py <case prefix>: scripts/prof_cfg.sh "c3 fixed B=512" name
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$2
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$2 -o p -- python $GRAFT_REPO_ROOT/scripts/time_configs.py "$1" > $GRAFT_REPO_ROOT/gpurun_out/$2_run.log 2>&1
F=$(find /tmp/prof_$2 -name '*kernel_stats.csv' | head -1)
cp "$F" $GRAFT_REPO_ROOT/gpurun_out/$2_kernel_stats.csv
