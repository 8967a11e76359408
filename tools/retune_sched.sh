O=gpurun_out/r3g; mkdir -p $O
for B in 4 1; do
  python -m poco_amd.tune --variant hrnet_w48_cls-cliff --batch $B --in-context --out $O/t.json >> $O/tune2.log 2>&1
  python -m poco_amd.tune --variant hrnet_w32-pare --batch $B --in-context --out $O/t.json >> $O/tune2.log 2>&1
done
for B in 16 4 1; do
  python -m poco_amd.tune --variant hrnet_w48_cls-cliff --batch $B --g3 --out $O/t.json >> $O/tune2.log 2>&1
  python -m poco_amd.tune --variant hrnet_w32-pare --batch $B --g3 --out $O/t.json >> $O/tune2.log 2>&1
done
cp poco_amd/tuned/gfx950.json $O/gfx950_2.json
val() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); sb=d.get('small_batch',{}); print(d['value'], d['ms_per_step'], sb.get('B1',{}).get('crops_per_s'), sb.get('B4',{}).get('crops_per_s'), sb.get('B16',{}).get('crops_per_s'))"; }
echo "w48: $(python bench.py --no-cpu-baseline --no-stream 2>&1 | val)" >> $O/after2.txt
echo "pare: $(python bench.py --variant hrnet_w32-pare --batch 32 --no-cpu-baseline --no-stream 2>&1 | val)" >> $O/after2.txt
