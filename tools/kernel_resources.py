#!/usr/bin/env python3
"""hipcc ... -Rpass-analysis=kernel-resource-usage 2>&1 | python tools/kernel_resources.py [filter]
One line per kernel: VGPRs, AGPRs, scratch, spills, occupancy, LDS."""
import re, subprocess, sys
pat = sys.argv[1] if len(sys.argv) > 1 else ""
rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"remark:\s+(.*?): (\S+) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
rows = [r for r in rows if pat in r["name"]]
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.split("\n")
for r, n in zip(rows, names):
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*\)$", "", n)
    print(f"{n[:70]:70s} V={r.get('VGPRs'):>3} A={r.get('AGPRs'):>3} sgprSpill={r.get('SGPRs Spill'):>3} vgprSpill={r.get('VGPRs Spill'):>4} scratch={r.get('ScratchSize [bytes/lane]'):>5} occ={r.get('Occupancy [waves/SIMD]')} lds={r.get('LDS Size [bytes/block]')}")
