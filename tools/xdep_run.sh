O=gpurun_out/r3d; mkdir -p $O; rm -f $O/ab.txt
val() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); sb=d.get('small_batch',{}); print(d['value'], d['ms_per_step'], sb.get('B1',{}).get('crops_per_s'), sb.get('B16',{}).get('crops_per_s'))"; }
python -m pytest tests/test_model_gpu.py -q -x -m gpu 2>&1 | tail -2 > $O/parity.log
for rep in 1 2 3; do for cfg in "POCO_NO_TAIL_LANES=1" "POCO_NO_TAIL_LANES=0"; do
  echo "$cfg w48: $(env $cfg python bench.py --no-cpu-baseline --no-stream 2>&1 | val)" >> $O/ab.txt
  echo "$cfg pare: $(env $cfg python bench.py --variant hrnet_w32-pare --batch 32 --no-cpu-baseline --no-stream 2>&1 | val)" >> $O/ab.txt
  echo "$cfg r50: $(env $cfg python bench.py --variant resnet50-cliff --no-cpu-baseline --no-stream 2>&1 | val)" >> $O/ab.txt
done; done
