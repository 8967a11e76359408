#!/bin/bash
# Round-end artifacts on the GPU box: bench lines of the three variants, rocprofv3 kernel stats (single lane = non-overlapping
# kernel durations, and the default 4 lanes) and the PMC passes (separate passes; never mixed with hip/hsa tracing).
# Usage (from the repo root on the box): tools/refresh_profiles.sh <tag> [pmc]   -> gpurun_out/<tag>/ ; copy what is to be
# judged into profiles/ (bench.py's roofline.traffic reads profiles/<tag>_pmc_*_summary.json and checks its kernel-source digest).
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
[ -n "$ONLY" ] || rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# ONLY=<short tag> (w48cliff | resnet50cliff | w32pare): re-take the kernel stats / PMC passes of one variant, keep the rest
[ -n "$ONLY" ] || {
python $R/bench.py 2>$OUT/bench_w48.err | tail -1 > $OUT/${TAG}_bench_hrnet_w48_cls-cliff.json
python $R/bench.py --variant resnet50-cliff --batch 64 --no-stream --no-side 2>/dev/null | tail -1 > $OUT/${TAG}_bench_resnet50-cliff.json
python $R/bench.py --variant hrnet_w32-pare --batch 32 --no-stream --no-side 2>/dev/null | tail -1 > $OUT/${TAG}_bench_hrnet_w32-pare.json
python $R/bench.py --no-graph --no-cpu-baseline --no-stream --no-side --no-variants 2>/dev/null | tail -1 > $OUT/${TAG}_bench_hrnet_w48_cls-cliff_nograph.json
}
prof_variant() {   # variant batch short-tag
  V=$1; B=$2; T=$3
  ARGS="--variant $V --batch $B --no-cpu-baseline --no-stream --no-dominant --no-graph --no-side --no-variants"
  for L in 1 4; do
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -o bench -- \
      python $R/bench.py --steps 10 --warmup 3 --lanes $L $ARGS > $OUT/ks_${T}_l$L.log 2>&1
    cp $(find $OUT/ks -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_${T}_b${B}_lanes${L}_kernel_stats.csv; rm -rf $OUT/ks
    [ "$T" = "w48cliff" ] || break      # 4-lane statistics only for the headline variant
  done
  if [ "$4" = "pmc" ]; then
    # PMC passes on ONE lane: with 4 lanes kernels of different branches share the chip and a kernel's GRBM_GUI_ACTIVE
    # (the time base of mfma_busy) counts the other kernels' time too
    CMD="python $R/bench.py --steps 3 --warmup 2 --lanes 1 $ARGS"
    timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES --output-format csv -d $OUT/p/p1 -o p1 -- $CMD > $OUT/p1.log 2>&1
    timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/p/p2 -o p2 -- $CMD > $OUT/p2.log 2>&1
    timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/p/p3 -o p3 -- $CMD > $OUT/p3.log 2>&1
    timeout 900 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $OUT/p/p4 -o p4 -- $CMD > $OUT/p4.log 2>&1
    python $R/tools/pmc_summary.py $OUT/p $OUT/${TAG}_pmc_${T}_b${B}_summary.json 5
    rm -rf $OUT/p
  fi
}
[ -n "$ONLY" ] && [ "$ONLY" != w48cliff ] || prof_variant hrnet_w48_cls-cliff 64 w48cliff $2
[ -n "$ONLY" ] && [ "$ONLY" != resnet50cliff ] || prof_variant resnet50-cliff 64 resnet50cliff $2
[ -n "$ONLY" ] && [ "$ONLY" != w32pare ] || prof_variant hrnet_w32-pare 32 w32pare $2
ls $OUT; cat $OUT/*bench*.json | cut -c1-300
