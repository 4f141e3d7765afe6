#!/bin/bash
# Register / spill summary of the MFMA kernels (hipcc resource-usage remarks, compacted).
cd "$(dirname "$0")/../muzero_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c net_mfma.hip -o /tmp/nm_regs.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re,sys
name=None
for l in sys.stdin:
    m=re.search(r"Function Name: (\S+)",l)
    if m:
        n=m.group(1)
        n=re.sub(r"_ZN2mz12_GLOBAL__N_1\d","",n)
        n=re.sub(r"EvNS_7NetDims.*","",n); name=n; vals={}
    for key in ("VGPRs","AGPRs","VGPRs Spill","ScratchSize [bytes/lane]","Occupancy [waves/SIMD]"):
        m=re.search(re.escape(key)+r": (\d+)",l)
        if m and name: vals[key]=m.group(1)
    if "LDS Size" in l and name:
        print(name[:70].ljust(70)," ".join(f"{k}={v}" for k,v in vals.items())); name=None
'
