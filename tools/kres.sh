#!/bin/bash
# register / scratch / occupancy figures of every kernel of one HIP source (compile-time remarks, no GPU):
#   tools/kres.sh sg_pr_amd/csrc/sgpr_embed.hip [extra hipcc flags]
src=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed "$@" -I$root/include -I$root/sg_pr_amd/csrc --cuda-device-only -c $src -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, sys
cur = None
for line in sys.stdin:
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        if cur: print(cur)
        cur = t.split(":", 1)[1].strip()[:60].ljust(60)
    elif any(t.startswith(k) for k in ("VGPRs:", "ScratchSize", "Occupancy", "LDS Size", "TotalSGPRs", "AGPRs")):
        cur += "  " + t.replace(" [bytes/lane]", "").replace(" [waves/SIMD]", "").replace(" [bytes/block]", "")
if cur: print(cur)
'
