O=gpurun_out/r3s2; mkdir -p $O
for B in 16 4 1; do
  python -m poco_amd.tune --variant hrnet_w48_cls-cliff --batch $B --small-planes --out $O/t.json >> $O/tune.log 2>&1
done
for B in 16 4 1; do
  python -m poco_amd.tune --variant hrnet_w32-pare --batch $B --small-planes --out $O/t.json >> $O/tune.log 2>&1
done
cp poco_amd/tuned/gfx950.json $O/gfx950.json
python bench.py --no-cpu-baseline --no-stream 2>/dev/null | tail -1 > $O/bench_w48.json
python bench.py --variant hrnet_w32-pare --batch 32 --no-cpu-baseline --no-stream 2>/dev/null | tail -1 > $O/bench_pare.json
