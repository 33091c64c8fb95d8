#!/bin/bash
# Full rocprofv3 evidence of one bench workload on the GPU box:  scripts/profile_workload.sh <c2|c3|c4|c5> <tag, e.g. r02> [extra bench.py flags, e.g. "--scaling strong"]
#   <tag>_<w>_kernel_stats.csv            --kernel-trace --stats summary
#   <tag>_<w>_bench_under_rocprof.json    the JSON line of that run
#   <tag>_<w>_pmc_hbm_traffic.json        FETCH_SIZE / WRITE_SIZE per dispatch (separate --pmc passes, --kernel-trace only)
#   <tag>_<w>_pmc_sq.json                 SQ instruction / MFMA counters per dispatch
# written under gpurun_out/; copy the ones to keep into profiles/.
W=$1; TAG=$2; EXTRA=$3; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export MMG_BENCH_MIN_SECONDS=${MMG_BENCH_MIN_SECONDS:-0.3}     # (short timed window: the traces stay small; the un-profiled bench uses 2 s)
CMD="python $R/bench.py --workload $W --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs $EXTRA"
rm -rf /tmp/ks /tmp/pf /tmp/pw /tmp/pc /tmp/pd
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o ks -- $CMD > $O/${TAG}_${W}_bench_under_rocprof.json 2> /tmp/ks.err
find /tmp/ks -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_${W}_kernel_stats.csv \;
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -o pf -- $CMD > /dev/null 2> /tmp/pf.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -o pw -- $CMD > /dev/null 2> /tmp/pw.err
python $R/scripts/pmc_summary.py /tmp/pf /tmp/pw $O/${TAG}_${W}_pmc_hbm_traffic.json > /dev/null
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d /tmp/pc -o pc -- $CMD > /dev/null 2> /tmp/pc.err
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY --kernel-trace --output-format csv -d /tmp/pd -o pd -- $CMD > /dev/null 2> /tmp/pd.err
python $R/scripts/pmc_counters.py $O/${TAG}_${W}_pmc_sq.json /tmp/pc /tmp/pd > /dev/null
for f in /tmp/ks.err /tmp/pf.err /tmp/pw.err /tmp/pc.err /tmp/pd.err; do tail -n 1 $f; done
