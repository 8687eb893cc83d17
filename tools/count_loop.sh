#!/bin/bash
# usage: count_loop.sh <mangled-kernel-substring>   (development aid: hot-loop instruction mix)
set -e
cd /root/repo/icp_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -ffp-contract=off -fPIC -S --cuda-device-only -o /tmp/icpx_tmp/capi.s icpx_capi.hip 2>/dev/null
python3 - "$1" <<'PY'
import re,sys
txt=open('/tmp/icpx_tmp/capi.s').read().split('\n')
key=sys.argv[1]
start=[i for i,l in enumerate(txt) if l.startswith('_ZN4icpx11pair_kernel'+key) and l.rstrip().endswith(':') or (l.startswith('_ZN4icpx11pair_kernel'+key) and ': ' in l)][0]
end=next(i for i in range(start,len(txt)) if 's_endpgm' in txt[i])
body=txt[start:end]
# split into labelled blocks
blocks={}; cur='entry'; blocks[cur]=[]
for l in body:
    m=re.match(r'^(\.LBB\d+_\d+):',l)
    if m: cur=m.group(1); blocks[cur]=[]
    else: blocks[cur].append(l.strip())
best=max(blocks.items(), key=lambda kv: sum('v_cmp_lt_f64' in x for x in kv[1]))
ins=[x.split()[0] for x in best[1] if x and not x.startswith(';') and not x.startswith('.')]
from collections import Counter
c=Counter(ins); ncmp=c['v_cmp_lt_f64_e32']+c.get('v_cmp_lt_f64_e64',0)+c.get('v_cmp_gt_f64_e32',0)
valu=sum(v for k,v in c.items() if k.startswith('v_'))
print(best[0],'cmps',ncmp,'valu',valu,'valu/eval',round(valu/max(ncmp,1),2)); print(c.most_common(12))
vg=[l for l in body if '.vgpr_count' in l or 'NumVgprs' in l or 'ScratchSize' in l]
print([l.strip() for l in txt[end:end+80] if 'NumVgprs' in l or 'ScratchSize' in l or 'Occupancy' in l])
PY
