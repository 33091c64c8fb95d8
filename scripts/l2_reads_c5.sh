#!/bin/bash
# L2 read requests of the forward conversation kernel at BASELINE config 5 (D = 1000, B = 2048, continuous), per-sample
# generic kernel (MMG_NO_TILE=1, the round-1 path) vs the sample-tile kernel: rocprofv3 --pmc TCC counters, --kernel-trace only.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
for mode in generic tile; do
  rm -rf /tmp/l2_$mode
  if [ $mode = generic ]; then export MMG_NO_TILE=1; else unset MMG_NO_TILE; fi
  timeout 300 rocprofv3 --pmc TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d /tmp/l2_$mode -o l2 -- python $R/scripts/time_configs.py c5 > /dev/null 2> /tmp/l2_$mode.err
  tail -n 1 /tmp/l2_$mode.err
done
unset MMG_NO_TILE
python $R/scripts/pmc_counters.py $O/r02_config5_l2_generic.json /tmp/l2_generic > /dev/null
python $R/scripts/pmc_counters.py $O/r02_config5_l2_tile.json /tmp/l2_tile > /dev/null
