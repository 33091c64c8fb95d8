# SQ counters of the bench workload, three passes (counters need their own runs: --kernel-trace only)
R=/root/repo; O=$R/gpurun_out/r01b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d /tmp/pa -o pa -- $CMD > /dev/null 2> /tmp/pa.err
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/pb -o pb -- $CMD > /dev/null 2> /tmp/pb.err
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d /tmp/pc -o pc -- $CMD > /dev/null 2> /tmp/pc.err
python $R/scripts/pmc_counters.py $O/pmc_sq_counters.json /tmp/pa /tmp/pb /tmp/pc
for f in /tmp/pa.err /tmp/pb.err /tmp/pc.err; do tail -n 2 $f; done
