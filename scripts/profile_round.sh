set -x
R=/root/repo; O=$R/gpurun_out/r01b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 python $R/bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o ks -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_under_rocprof.json 2> /tmp/ks.err
find /tmp/ks -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
head -12 $O/kernel_stats.csv
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -o pf -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2> /tmp/pf.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -o pw -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2> /tmp/pw.err
python $R/scripts/pmc_summary.py /tmp/pf /tmp/pw $O/pmc_hbm_traffic.json | head -60
ls /tmp/pf /tmp/pf/* | head
