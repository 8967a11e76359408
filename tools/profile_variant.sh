#!/bin/bash
# rocprofv3 kernel stats (single lane) + PMC passes for one variant: tools/profile_variant.sh <variant> <batch> <tag>
V=$1; B=$2; TAG=$3
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--variant $V --batch $B --no-cpu-baseline --no-dominant --no-graph"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -o bench -- python $R/bench.py --steps 10 --warmup 3 --lanes 1 $ARGS > $OUT/ks.log 2>&1
cp $(find $OUT/ks -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_lanes1_kernel_stats.csv; rm -rf $OUT/ks
CMD="python $R/bench.py --steps 3 --warmup 2 $ARGS"
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -o p1 -- $CMD > $OUT/p1.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/p2 -o p2 -- $CMD > $OUT/p2.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/p3 -o p3 -- $CMD > $OUT/p3.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $OUT/p4 -o p4 -- $CMD > $OUT/p4.log 2>&1
python $R/tools/pmc_summary.py $OUT $OUT/${TAG}_pmc_summary.json > /dev/null
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4
ls $OUT
