#!/bin/bash
# rocprofv3 kernel-trace summary of one bench workload: scripts/prof_config.sh <workload> <out-name> [steps]
# writes gpurun_out/<out-name>_kernel_stats.csv (copy the ones to keep into profiles/)
W=$1; NAME=$2; STEPS=${3:-50}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$NAME
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$NAME -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $W --steps $STEPS --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${NAME}_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/${NAME}_prof.log
F=$(find /tmp/prof_$NAME -name '*kernel_stats.csv' | head -1)
cp "$F" $GRAFT_REPO_ROOT/gpurun_out/${NAME}_kernel_stats.csv
