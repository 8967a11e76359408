O=gpurun_out/r3k; mkdir -p $O
for v in 1 0; do
  POCO_NO_KMERGE=$v python bench.py --no-cpu-baseline --no-stream > $O/full_$v.json 2>/dev/null
done
python -m poco_amd.tune --variant hrnet_w48_cls-cliff --batch 64 --g3 --out $O/t2.json > $O/tune_g3.log 2>&1
python -m poco_amd.tune --variant hrnet_w32-pare --batch 32 --g3 --out $O/t2.json >> $O/tune_g3.log 2>&1
cp poco_amd/tuned/gfx950.json $O/gfx950_g3.json
for rep in 1 2; do
  echo "w48: $(python bench.py --no-side --no-cpu-baseline --no-stream 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")" >> $O/ab2.txt
  echo "pare: $(python bench.py --variant hrnet_w32-pare --batch 32 --no-side --no-cpu-baseline --no-stream 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")" >> $O/ab2.txt
done
