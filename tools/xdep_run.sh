O=gpurun_out/r3d; mkdir -p $O; rm -f $O/ab.txt
val() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('small_batch',{}).get('B16',{}).get('crops_per_s'))"; }
for rep in 1 2 3; do for cfg in "POCO_T1_OPEN=0" "POCO_T1_OPEN=1"; do
  echo "$cfg w48: $(env $cfg python bench.py --no-cpu-baseline --no-stream 2>&1 | val)" >> $O/ab.txt
  echo "$cfg pare: $(env $cfg python bench.py --variant hrnet_w32-pare --batch 32 --no-side --no-cpu-baseline --no-stream 2>&1 | val)" >> $O/ab.txt
done; done
