#!/bin/bash
# L2 read requests of the forward conversation at BASELINE config 5's per-GPU shard (D = 1000, 256 samples, continuous):
# per-sample generic kernel (MMG_NO_MC=1, the round-1/2 path) vs k_conversation_mc.  rocprofv3 --pmc TCC counters, --kernel-trace only.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; TAG=${1:-r03}; cd /tmp && export TMPDIR=/tmp
export MMG_BENCH_MIN_SECONDS=0.2
for mode in generic mc; do
  rm -rf /tmp/l2_$mode
  if [ $mode = generic ]; then export MMG_NO_MC=1; else unset MMG_NO_MC; fi
  timeout 300 rocprofv3 --pmc TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d /tmp/l2_$mode -o l2 -- python $R/bench.py --workload c5 --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs > /dev/null 2> /tmp/l2_$mode.err
  tail -n 1 /tmp/l2_$mode.err
done
unset MMG_NO_MC
python $R/scripts/pmc_counters.py $O/${TAG}_config5_l2_generic.json /tmp/l2_generic > /dev/null
python $R/scripts/pmc_counters.py $O/${TAG}_config5_l2_mc.json /tmp/l2_mc > /dev/null
