#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r1b
mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/w48_l1 -o bench -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-dominant --lanes 1 > $OUT/w48_l1.log 2>&1
rm -f $OUT/w48_l1/*kernel_trace.csv
tail -1 $OUT/w48_l1.log | cut -c1-300
