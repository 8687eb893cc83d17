#!/bin/bash
# development aid: VGPR / spill / scratch numbers of the fused kernels in the built objects
for o in icp_amd/lib/pair_d*.o; do
  t=$(mktemp -d)
  /opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin $o $t/fat.bin
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$t/fat.bin --output=$t/co --unbundle
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes $t/co > $t/notes.txt
  python3 - $t/notes.txt $o <<'PY'
import re,sys
txt=open(sys.argv[1]).read()
for e in re.split(r'\n\s+- \.agpr_count', txt)[1:]:
    nm=re.search(r'\.name:\s+(\S+)',e); v=re.search(r'\.vgpr_count:\s+(\d+)',e); sp=re.search(r'\.vgpr_spill_count:\s+(\d+)',e); pr=re.search(r'\.private_segment_fixed_size:\s+(\d+)',e)
    if nm and 'pair_kernel' in nm.group(1):
        k=re.search(r'pair_kernelILi(\d)ELi(\d)ELi(\d)ELi(\d+)ELb(\d)',nm.group(1))
        print(sys.argv[2].split('/')[-1], 'D%s M%s QPT%s T%s P%s'%k.groups(), 'vgpr',v.group(1),'spill',sp.group(1),'scratch',pr.group(1))
PY
  rm -rf $t
done
