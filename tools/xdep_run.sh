O=gpurun_out/r3d; mkdir -p $O; rm -f $O/ab.txt
val() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); sb=d.get('small_batch',{}); print(d['value'], d['ms_per_step'], sb.get('B1',{}).get('crops_per_s'), sb.get('B16',{}).get('crops_per_s'))"; }
python -m pytest tests/test_model_gpu.py -q -x -m gpu 2>&1 | tail -2 > $O/parity.log
python tools/stress.py > $O/stress.log 2>&1
python tools/stress.py hrnet_w32-pare 32 100 >> $O/stress.log 2>&1
for rep in 1 2; do
  echo "w48: $(python bench.py --no-cpu-baseline --no-stream 2>&1 | val)" >> $O/ab.txt
  echo "pare: $(python bench.py --variant hrnet_w32-pare --batch 32 --no-cpu-baseline --no-stream 2>&1 | val)" >> $O/ab.txt
done
