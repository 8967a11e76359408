#!/bin/bash
# usage: tools/regs.sh poco_amd/csrc/file.hip  -> terse per-kernel register/scratch summary
hipcc -O3 -std=c++17 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage -c "$1" -o /tmp/regs_tmp.o 2>&1 | python3 -c "
import sys,re
rows=[];cur=None
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur={'n':m.group(1)}; rows.append(cur)
    for k in ['VGPRs','ScratchSize','Occupancy','LDS Size']:
        m=re.search(r'\s'+re.escape(k)+r'[^:]*: (\d+)',l)
        if m and cur is not None: cur.setdefault(k,m.group(1))
for r in rows:
    m=re.search(r'(\w+?)I((?:Li\d+E)+)E',r['n'])
    nm = (m.group(1)[-20:]+' '+','.join(re.findall(r'Li(\d+)E',m.group(2)))) if m else r['n'][-40:]
    print(nm, 'V=%s S=%s occ=%s'%(r.get('VGPRs'),r.get('ScratchSize'),r.get('Occupancy')))
"
