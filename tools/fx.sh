for dbg in 0 1; do echo "DBG=$dbg"; POCO_CONV_DBG=$dbg python tools/wino_slope.py 2>&1 | grep ", 4)" | cut -c1-200; done
