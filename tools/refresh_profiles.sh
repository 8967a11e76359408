#!/bin/bash
# Round-end artifacts on the GPU box: bench lines of the three variants, rocprofv3 kernel stats
# (single lane = non-overlapping kernel durations, and the default 4 lanes) and the PMC passes.
# Usage (from the repo root on the box): tools/refresh_profiles.sh <tag>   -> gpurun_out/<tag>/
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 30 --warmup 5 2>$OUT/bench_w48.err | tail -1 > $OUT/${TAG}_bench_hrnet_w48_cls-cliff.json
python $R/bench.py --steps 30 --warmup 5 --variant resnet50-cliff --batch 64 2>/dev/null | tail -1 > $OUT/${TAG}_bench_resnet50-cliff.json
python $R/bench.py --steps 30 --warmup 5 --variant hrnet_w32-pare --batch 32 2>/dev/null | tail -1 > $OUT/${TAG}_bench_hrnet_w32-pare.json
python $R/bench.py --steps 30 --warmup 5 --no-graph --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/${TAG}_bench_hrnet_w48_cls-cliff_nograph.json
for L in 1 4; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks_l$L -o bench -- \
    python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-dominant --no-graph --lanes $L > $OUT/ks_l$L.log 2>&1
  cp $OUT/ks_l$L/*/*kernel_stats.csv $OUT/${TAG}_bench_w48cliff_b64_lanes${L}_kernel_stats.csv 2>/dev/null || \
    cp $OUT/ks_l$L/*kernel_stats.csv $OUT/${TAG}_bench_w48cliff_b64_lanes${L}_kernel_stats.csv
  rm -rf $OUT/ks_l$L
done
if [ "$2" = "pmc" ]; then
  BENCH_ARGS="--no-graph" bash $R/tools/run_pmc.sh > $OUT/pmc.log 2>&1
  cp $R/gpurun_out/pmc_r1/summary.json $OUT/${TAG}_pmc_w48cliff_b64_summary.json
fi
cat $OUT/*.json | cut -c1-400
