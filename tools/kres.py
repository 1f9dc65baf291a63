#!/usr/bin/env python3
"""tools/kres.py RESOURCES [PATTERN] -- one line per kernel from hipcc's -Rpass-analysis=kernel-resource-usage output:
name, VGPRs, AGPRs, SGPRs, spills, scratch, LDS, occupancy."""
import re
import sys

text = open(sys.argv[1], errors="replace").read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
cur = None
rows = []
for line in text.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
for r in rows:
    if pat in r["name"]:
        print("%-90s vgpr %3d agpr %3d sgpr %3d spill %d/%d scratch %d occ %d" % (
            r["name"][:90], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("TotalSGPRs", -1), r.get("VGPRs Spill", -1),
            r.get("SGPRs Spill", -1), r.get("ScratchSize", -1), r.get("Occupancy", -1)))
