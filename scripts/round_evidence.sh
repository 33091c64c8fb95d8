#!/bin/bash
# End-of-round evidence on the GPU box, written under gpurun_out/<tag>_*: the full GPU suite (+ its max-error table), the
# un-profiled bench line, rocprofv3 tables + HBM / SQ counters of every workload, the DP overhead and the in-kernel timelines.
#   scripts/round_evidence.sh r04
TAG=${1:-rXX}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/${TAG}_gpu_tests.log; cp tests/out/parity_maxerr.json $O/${TAG}_parity_maxerr.json 2>/dev/null
python bench.py > $O/${TAG}_bench_line.json 2> /dev/null
for w in c2 c3 c4 c4r256 c5; do bash scripts/profile_workload.sh $w $TAG; done
bash scripts/profile_workload.sh c3 ${TAG}_strong "--scaling strong"
bash scripts/profile_workload.sh c5 ${TAG}_strong "--scaling strong"
timeout 300 python scripts/dp_overhead.py 2>&1 | grep -v "^/opt\|^\[W" > $O/${TAG}_dp_overhead.log
FIXED=0 timeout 300 python scripts/timeline.py 2>&1 | grep -v "^/opt\|hipcc" > $O/${TAG}_timeline_c2.log
timeout 300 python scripts/mc_timeline.py 2>&1 | grep -v "^/opt\|hipcc" > $O/${TAG}_mc_timeline_c5.log
timeout 300 python scripts/rc_timeline.py 2>&1 | grep -v "^/opt\|hipcc" > $O/${TAG}_rc_timeline_c4r256.log
python scripts/time_configs.py 2>&1 | grep -v "^/opt" > $O/${TAG}_time_configs.log
tail -3 $O/${TAG}_gpu_tests.log; cat $O/${TAG}_time_configs.log
