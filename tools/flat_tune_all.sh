#!/bin/bash
# in-context flat-item pass for every (variant, batch) the tuned table serves; the table is written in place and copied to gpurun_out/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/flat_tune; mkdir -p $O
for vb in "hrnet_w48_cls-cliff 64" "hrnet_w48_cls-cliff 128" "hrnet_w32-pare 32" "hrnet_w48_cls-cliff 32" "hrnet_w48_cls-cliff 16" "resnet50-cliff 64" "hrnet_w32-pare 16" "hrnet_w32-pare 64"; do
  set -- $vb
  python $R/tools/flat_tune.py $1 $2 --two-pass --write > $O/$1_b$2.log 2>&1
  tail -2 $O/$1_b$2.log
done
cp $R/poco_amd/tuned/gfx950.json $O/gfx950.json
