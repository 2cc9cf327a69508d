#!/bin/bash
# usage: tools/kres.sh file.hip  -> kernel name, VGPRs, spills, occupancy, LDS
hipcc --offload-arch=gfx950 -O3 -std=c++17 -c "$1" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
cur=None
for line in sys.stdin:
    m=re.search(r'remark:\s+(.*?)\s*\[-Rpass', line)
    if not m: continue
    t=m.group(1)
    if t.startswith('Function Name'):
        cur=t.split(':',1)[1].strip(); print(); print(cur[:150], end=' | ')
    elif any(t.startswith(k) for k in ('VGPRs:','AGPRs','VGPR Spill','Occupancy','LDS Size','ScratchSize','SGPRs:')):
        print(t.replace(' [waves/SIMD]','').replace(' [bytes/block]','').replace(' [bytes/lane]',''), end='; ')
print()
"
