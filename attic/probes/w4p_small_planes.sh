O=gpurun_out/r3t; mkdir -p $O
: > $O/w4p_7x7.txt
for B in 1 4 16 64; do
 for NT in 1 2 3; do for NI in 1 2 4 8; do
  echo "B=$B 7x7 384 NT=$NT NI=$NI: $(python tools/conv_time.py $B 7 7 384 384 3 1 1 $NT 2 4 8 $NI 8 2>&1 | tail -1)" >> $O/w4p_7x7.txt
 done; done
done
for B in 1 4 16; do
 for NT in 1 2 3; do for NI in 1 2; do
  echo "B=$B 14x14 192 NT=$NT NI=$NI: $(python tools/conv_time.py $B 14 14 192 192 3 1 1 $NT 2 4 16 $NI 8 2>&1 | tail -1)" >> $O/w4p_7x7.txt
 done; done
done
