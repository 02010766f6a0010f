#!/bin/bash
# Per-kernel register / spill / scratch table of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
# usage: tools/kernel_resources.sh vidil_amd/csrc/gemmpp.hip [extra hipcc flags]
f=$1; shift
cd "$(dirname "$f")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -c "$(basename "$f")" -o /tmp/kres_probe.o \
  -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c '
import re, sys, subprocess
rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs|ScratchSize|NumSgprs): (\S+)", line)
    if not m: continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    elif cur is not None: cur[k] = v
for r in rows:
    try: name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    except Exception: name = r["name"]
    name = re.sub(r"\(anonymous namespace\)::", "", name).replace("(vidil_gemm_args)", "")
    print("%-90s VGPR %3s AGPR %3s SGPR %3s spillV %3s spillS %3s scratch %4s occ %s" % (name[:90], r.get("VGPRs"), r.get("AGPRs"), r.get("SGPRs"), r.get("VGPRs Spill"), r.get("SGPRs Spill"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]")))
'
