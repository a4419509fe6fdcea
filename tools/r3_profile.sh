#!/bin/bash
# round-3 profiling run on the GPU box: rocprofv3 kernel stats of bench.py, the two HBM-traffic PMC passes, and SQ / LDS counters of
# the two Winograd kernels on one layer.  Outputs under gpurun_out/r3_prof/ (summaries are copied to profiles/ by hand).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3_prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err
rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats -o r3 -- python3 $R/bench.py --steps 7 --warmup 2 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/stats.err
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/fetch -o f -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > /dev/null 2> $O/fetch.err
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/write -o w -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > /dev/null 2> $O/write.err
export WINO_CHECK_F44=1
rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES -d $O/w44_p1 -o p -- python3 $R/tools/wino_check.py one 256 256 128 128 32 > $O/w44_p1.log 2>&1
rocprofv3 --output-format csv --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM -d $O/w44_p2 -o p -- python3 $R/tools/wino_check.py one 256 256 128 128 32 > $O/w44_p2.log 2>&1
export WINO_CHECK_F44=0
rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES -d $O/w22_p1 -o p -- python3 $R/tools/wino_check.py one 256 256 128 128 32 > $O/w22_p1.log 2>&1
rocprofv3 --output-format csv --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM -d $O/w22_p2 -o p -- python3 $R/tools/wino_check.py one 256 256 128 128 32 > $O/w22_p2.log 2>&1
cd $O
find . -name "*kernel_stats.csv" | head -3
for d in w44_p1 w44_p2 w22_p1 w22_p2; do f=$(find $d -name "*counter_collection.csv" | head -1); echo "== $d"; python3 $R/tools/pmc_kernel.py $f wino > $O/$d.txt 2>&1; cat $O/$d.txt; done
ff=$(find fetch -name "*counter_collection.csv" | head -1); fw=$(find write -name "*counter_collection.csv" | head -1)
python3 $R/tools/pmc_summary.py $ff $fw $O/pmc_traffic.json
# keep only the small summaries (the raw traces are large)
find . -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
rm -rf stats/*/*kernel_trace* fetch write w44_p1 w44_p2 w22_p1 w22_p2 2>/dev/null
du -sh $O
