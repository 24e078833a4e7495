#!/bin/bash
# Register / LDS / occupancy table of one HIP source's gfx950 kernels (compiler's resource-usage remarks):
#   tools/kernel_regs.sh conv3x3_dma.hip [extra hipcc flags]
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -I/root/repo/include -I/root/repo/combat_amd/csrc \
  --cuda-device-only -c /root/repo/combat_amd/csrc/$src -o /tmp/_regs.co -Rpass-analysis=kernel-resource-usage "$@" 2>&1 |
python3 -c '
import re,sys
cur={}
for l in sys.stdin:
    m=re.search(r"remark: [^:]*:\d+:\d+: +(.*?): +(.*?) *\[-Rpass",l) or re.search(r"remark: +(Function Name|[A-Za-z ]+): +(\S+)",l)
    m=re.search(r"remark:\s+(Function Name|Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)",l)
    if not m: continue
    k,v=m.groups()
    if k in ("Function Name","Name"):
        cur={"name":v}
    cur[k]=v
    if k.startswith("LDS"):
        import subprocess
        n=subprocess.run(["c++filt",cur["name"]],capture_output=True,text=True).stdout.strip()
        n=n.replace("(anonymous namespace)::","").split("(")[0]
        print("%-62s vgpr %3s agpr %3s sgpr %3s scratch %4s occ %s spill v%s s%s"%(n[:62],cur.get("VGPRs"),cur.get("AGPRs"),cur.get("TotalSGPRs"),cur.get("ScratchSize [bytes/lane]"),cur.get("Occupancy [waves/SIMD]"),cur.get("VGPRs Spill"),cur.get("SGPRs Spill")))
'
