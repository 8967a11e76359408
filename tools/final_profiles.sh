#!/bin/bash
# round-end artifacts in one go: tools/refresh_profiles.sh <tag> pmc, then the bench lines re-run with the fresh PMC summaries in place
# (bench.py reads profiles/<tag>_pmc_*_summary.json), everything under gpurun_out/<tag>/ ; copy to profiles/ afterwards.
TAG=${1:-r04}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG
bash $R/tools/refresh_profiles.sh $TAG pmc > $R/gpurun_out/refresh_$TAG.log 2>&1
cp $O/${TAG}_pmc_*_summary.json $R/profiles/
cd /tmp
python $R/bench.py --no-stream --no-side --no-variants 2>/dev/null | tail -1 > $O/${TAG}_bench_hrnet_w48_cls-cliff.json
python $R/bench.py --variant resnet50-cliff --batch 64 --no-stream --no-side 2>/dev/null | tail -1 > $O/${TAG}_bench_resnet50-cliff.json
python $R/bench.py --variant hrnet_w32-pare --batch 32 --no-stream --no-side 2>/dev/null | tail -1 > $O/${TAG}_bench_hrnet_w32-pare.json
python $R/bench.py --batch 128 --no-stream --no-side --no-variants --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench_hrnet_w48_cls-cliff_b128.json
python $R/bench.py 2>/dev/null | tail -1 > $O/${TAG}_bench_default_line.json
cut -c1-120 $O/${TAG}_bench_*.json
