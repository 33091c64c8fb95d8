#!/bin/bash
# per-kernel durations and the idle gap before each kernel of the configs[1] minibatch (rocprofv3 kernel trace, rocpd database)
#   scripts/gaps_c2.sh tag [VAR=1 ...]
cd /tmp && export TMPDIR=/tmp
tag=$1; shift
rm -rf /tmp/gaps_$tag
env "$@" MMG_BENCH_MIN_SECONDS=0.3 rocprofv3 --kernel-trace --output-format rocpd -d /tmp/gaps_$tag -o p -- python $GRAFT_REPO_ROOT/bench.py --workload ${W:-c2} --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs --no-cli > /dev/null 2>&1
F=$(find /tmp/gaps_$tag -name '*.db' | head -1)
echo "== $tag $@"
python $GRAFT_REPO_ROOT/scripts/rocpd_gaps.py "$F" | head -9
