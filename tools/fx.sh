for dbg in 0 1; do echo "DBG=$dbg"; POCO_CONV_DBG=$dbg timeout 600 python tools/conv_fixed_cost.py 2>&1 | grep -v amdgpu | grep "1, 1)\|2, 1)" | head -4 | cut -c1-150; done
