O=$GRAFT_REPO_ROOT/gpurun_out/tt; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in hrnet_w48_cls-cliff resnet50-cliff; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt_$v -o t -- python $R/bench.py --variant $v --batch 64 --steps 6 --warmup 3 --no-cpu-baseline --no-stream --no-side --no-dominant --no-variants > $O/kt_$v.log 2>&1
  f=$(find $O/kt_$v -name "*kernel_trace.csv" | head -1)
  python $R/tools/tail_trace.py $f 50 > $O/tail_$v.txt 2>&1
  rm -rf $O/kt_$v
done
