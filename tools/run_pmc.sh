#!/bin/bash
# PMC passes of bench.py (separate passes: counters do not all fit in one; never mixed with hip/hsa tracing)
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_r1
rm -rf $OUT; mkdir -p $OUT
CMD="python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-dominant ${BENCH_ARGS}"
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -o p1 -- $CMD > $OUT/p1.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/p2 -o p2 -- $CMD > $OUT/p2.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/p3 -o p3 -- $CMD > $OUT/p3.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $OUT/p4 -o p4 -- $CMD > $OUT/p4.log 2>&1
ls -R $OUT | head -30
python $R/tools/pmc_summary.py $OUT $OUT/summary.json
tail -3 $OUT/p1.log | cut -c1-300
rm -f $OUT/*/*kernel_trace.csv   # large
