O=gpurun_out/r3d; mkdir -p $O; rm -f $O/ab.txt
val() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('small_batch',{}).get('B16',{}).get('crops_per_s'))"; }
python -m pytest tests/test_model_gpu.py -q -x -m gpu -k "pare" 2>&1 | tail -2 > $O/parity.log
for rep in 1 2 3; do for cfg in "POCO_NO_UP_LANES=1" "POCO_NO_UP_LANES=0"; do
  echo "$cfg pare: $(env $cfg python bench.py --variant hrnet_w32-pare --batch 32 --no-cpu-baseline --no-stream 2>&1 | val)" >> $O/ab.txt
done; done
