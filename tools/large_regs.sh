#!/bin/bash
# development aid: VGPR / spill / scratch numbers of the multi-kernel path's kernels (name filter: $1, default lg_nn)
pat=${1:-lg_nn}
o=${2:-icp_amd/lib/icpx_large.o}
t=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin $o $t/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$t/fat.bin --output=$t/co --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $t/co > $t/notes.txt
python3 - $t/notes.txt "$pat" <<'PY'
import re,sys,subprocess
txt=open(sys.argv[1]).read()
for e in re.split(r'\n\s+- \.agpr_count', txt)[1:]:
    nm=re.search(r'\.name:\s+(\S+)',e); v=re.search(r'\.vgpr_count:\s+(\d+)',e); sp=re.search(r'\.vgpr_spill_count:\s+(\d+)',e); pr=re.search(r'\.private_segment_fixed_size:\s+(\d+)',e)
    if nm and sys.argv[2] in nm.group(1):
        d=subprocess.run(['c++filt',nm.group(1)],capture_output=True,text=True).stdout.strip()
        print(d[:110], 'vgpr',v.group(1),'spill',sp.group(1),'scratch',pr.group(1))
PY
rm -rf $t
