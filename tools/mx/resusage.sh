#!/bin/bash
# per-kernel register / spill / scratch summary of one .hip source (hipcc -Rpass-analysis=kernel-resource-usage): tools/mx/resusage.sh csrc/tdnn_mxp.hip [extra flags]
src=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Rpass-analysis=kernel-resource-usage "$@" -c "$src" -o /tmp/resusage_$$.o 2>&1 | \
  python3 -c '
import re, sys
cur = None
rows = []
for line in sys.stdin:
    if "error" in line or "warning" in line: print(line.rstrip())
    m = re.search(r"remark:\s+(.*?): (.*?) \[-Rpass", line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name": cur = {"name": v}; rows.append(cur)
    elif cur is not None: cur[k] = v
for r in rows:
    print(r["name"][:60].ljust(60), "sgpr", r.get("TotalSGPRs"), "vgpr", r.get("VGPRs"), "agpr", r.get("AGPRs"), "scratch", r.get("ScratchSize [bytes/lane]"), "sspill", r.get("SGPRs Spill"), "vspill", r.get("VGPRs Spill"))
'
rm -f /tmp/resusage_$$.o
