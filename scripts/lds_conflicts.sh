#!/bin/bash
# LDS bank-conflict cycles per kernel of one bench workload (rocprofv3 --pmc, --kernel-trace only): scripts/lds_conflicts.sh <c2|c3|c4|c4r256|c5> <tag> [strong]
W=$1; TAG=$2; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O
EXTRA=""; [ -n "$3" ] && EXTRA="--scaling strong"
cd /tmp && export TMPDIR=/tmp MMG_BENCH_MIN_SECONDS=0.2
rm -rf /tmp/lc
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/lc -o lc -- python $R/bench.py --workload $W --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --no-cli $EXTRA > /dev/null 2> /tmp/lc.err
python $R/scripts/pmc_counters.py $O/${TAG}_${W}${3:+_strong}_lds_conflicts.json /tmp/lc | tail -40
tail -n 2 /tmp/lc.err
