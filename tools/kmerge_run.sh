# tune the shapes the K-merged fuse convs add, parity, A/B against POCO_NO_KMERGE=1 (one box)
O=gpurun_out/r3k; mkdir -p $O
python -m poco_amd.tune --variant hrnet_w48_cls-cliff --batch 64 128 32 16 4 1 --only-missing --out $O/t.json > $O/tune.log 2>&1
python -m poco_amd.tune --variant hrnet_w32-pare --batch 32 64 128 16 4 1 --only-missing --out $O/t.json >> $O/tune.log 2>&1
cp poco_amd/tuned/gfx950.json $O/gfx950.json
python -m pytest tests/test_model_gpu.py -q -x -m gpu 2>&1 | tail -4 > $O/parity.log
for rep in 1 2 3; do for v in 1 0; do
  echo "NO_KMERGE=$v w48: $(POCO_NO_KMERGE=$v python bench.py --no-side --no-cpu-baseline --no-stream 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")" >> $O/ab.txt
  echo "NO_KMERGE=$v pare: $(POCO_NO_KMERGE=$v python bench.py --variant hrnet_w32-pare --batch 32 --no-side --no-cpu-baseline --no-stream 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")" >> $O/ab.txt
done; done
