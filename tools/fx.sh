for dbg in 0 16; do echo "DBG=$dbg"; POCO_CONV_DBG=$dbg python tools/wino_slope.py 2>&1 | grep "4, 3, 4, 1, 3\|4, 1, 14, 1, 3" | cut -c1-200; done
python tools/wino_probe.py 2>&1 | grep -A2 "wino cand"
