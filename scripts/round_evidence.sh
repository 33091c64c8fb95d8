#!/bin/bash
# Evidence of a round on the GPU box, written under gpurun_out/ with the names profiles/ uses (copy what is to be judged there).
#   scripts/round_evidence.sh r05                 everything: GPU suite (+ max-error table), the un-profiled bench line, rocprofv3
#                                                 tables + HBM / SQ counters of every workload, DP overhead, in-kernel timelines
#   scripts/round_evidence.sh r05 c2              the rocprofv3 set of ONE bench workload (c2 | c3 | c4 | c4r256 | c5)
#   scripts/round_evidence.sh r05 c3 strong       ... of the whole global batch on this GPU (bench.py --scaling strong)
# Per workload (config<N> = c<N>; `strong_` prefix for --scaling strong):
#   <tag>_config<N>_kernel_stats.csv            rocprofv3 --kernel-trace --stats summary of `bench.py --workload c<N>`
#   <tag>_config<N>_bench_under_rocprof.json    the JSON line of that run
#   <tag>_config<N>_pmc_hbm_traffic.json        FETCH_SIZE / WRITE_SIZE per dispatch (separate --pmc passes, --kernel-trace only;
#                                               gfx950 correction as MI355X_MICROARCH.md prescribes: scripts/pmc_summary.py)
#   <tag>_config<N>_pmc_sq.json                 SQ instruction / MFMA / wait counters per dispatch (scripts/pmc_counters.py)
TAG=${1:-rXX}; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O

profile_workload() {   # <workload> <"" | strong>
  local W=$1 STRONG=$2 EXTRA="" NAME
  NAME=${TAG}_$([ -n "$STRONG" ] && echo strong_)config${W#c}
  [ -n "$STRONG" ] && EXTRA="--scaling strong"
  ( cd /tmp && export TMPDIR=/tmp
    export MMG_BENCH_MIN_SECONDS=${MMG_BENCH_MIN_SECONDS:-0.3}     # (short timed window: the traces stay small; the un-profiled bench uses 2 s)
    CMD="python $R/bench.py --workload $W --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --no-cli $EXTRA"
    rm -rf /tmp/ks /tmp/pf /tmp/pw /tmp/pc /tmp/pd
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o ks -- $CMD > $O/${NAME}_bench_under_rocprof.json 2> /tmp/ks.err
    find /tmp/ks -name "*kernel_stats.csv" -exec cp {} $O/${NAME}_kernel_stats.csv \;
    timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -o pf -- $CMD > /dev/null 2> /tmp/pf.err
    timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -o pw -- $CMD > /dev/null 2> /tmp/pw.err
    python $R/scripts/pmc_summary.py /tmp/pf /tmp/pw $O/${NAME}_pmc_hbm_traffic.json > /dev/null
    timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d /tmp/pc -o pc -- $CMD > /dev/null 2> /tmp/pc.err
    timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY --kernel-trace --output-format csv -d /tmp/pd -o pd -- $CMD > /dev/null 2> /tmp/pd.err
    python $R/scripts/pmc_counters.py $O/${NAME}_pmc_sq.json /tmp/pc /tmp/pd > /dev/null
    for f in /tmp/ks.err /tmp/pf.err /tmp/pw.err /tmp/pc.err /tmp/pd.err; do tail -n 1 $f; done )
}

if [ -n "$2" ]; then profile_workload $2 $3; exit 0; fi
cd $R
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/${TAG}_gpu_tests.log; cp tests/out/parity_maxerr.json $O/${TAG}_parity_maxerr.json 2>/dev/null
for w in c2 c3 c4 c4r256 c5; do profile_workload $w; done
profile_workload c3 strong
profile_workload c5 strong
# the un-profiled line AFTER the profiles: bench.py names its dominant kernel from the newest summary under profiles/, so the
# caller copies this run's *_kernel_stats.csv / *_pmc_hbm_traffic.json there first when the line is to cite them
python bench.py > $O/${TAG}_bench_line.json 2> /dev/null
timeout 300 python scripts/dp_overhead.py 2>&1 | grep -v "^/opt\|^\[W" > $O/${TAG}_dp_overhead.log
timeout 300 python scripts/game_timeline.py 2>&1 | grep -v "^/opt\|hipcc" > $O/${TAG}_game_timeline_c2.log
timeout 300 python scripts/tile_timeline.py c4 2>&1 | grep -v "^/opt\|hipcc\|Warning\|print(" > $O/${TAG}_tile_timeline_c4.log
timeout 300 python scripts/c4_ll_ab.py 40 > $O/${TAG}_c4_handoff_ab.log 2>&1
timeout 900 python scripts/path_ab.py 4 > $O/${TAG}_path_ab.log 2>&1
timeout 300 python scripts/mc_timeline.py 2>&1 | grep -v "^/opt\|hipcc" > $O/${TAG}_mc_timeline_c5.log
timeout 300 python scripts/rc_timeline.py 2>&1 | grep -v "^/opt\|hipcc" > $O/${TAG}_rc_timeline_c4r256.log
python scripts/time_configs.py 2>&1 | grep -v "^/opt" > $O/${TAG}_time_configs.log
tail -3 $O/${TAG}_gpu_tests.log; cat $O/${TAG}_time_configs.log
