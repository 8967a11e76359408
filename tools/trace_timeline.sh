# kernel trace of bench.py at batch $B (default 16), graph and eager, analysed by tools/timeline.py -> gpurun_out/r3t/timeline_*.txt
O=$GRAFT_REPO_ROOT/gpurun_out/r3t; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for mode in graph nograph; do
  EXTRA=""; [ $mode = nograph ] && EXTRA="--no-graph"
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/kt_$mode -o t -- python $R/bench.py --batch ${B:-16} --steps 6 --warmup 3 --no-cpu-baseline --no-stream --no-side --no-dominant --no-variants $EXTRA > $O/kt_$mode.log 2>&1
  f=$(find $O/kt_$mode -name "*kernel_trace.csv" | head -1)
  python $R/tools/timeline.py $f > $O/timeline_${mode}_b${B:-16}.txt 2>&1
  rm -rf $O/kt_$mode
done
