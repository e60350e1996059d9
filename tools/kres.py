#!/usr/bin/env python3
"""Compact per-kernel resource table from `hipcc -Rpass-analysis=kernel-resource-usage`."""
import re, subprocess, sys
flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
         "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null"]
for src in sys.argv[1:]:
    out = subprocess.run(["/opt/rocm/bin/hipcc", *flags, src], capture_output=True, text=True).stderr
    cur = {}
    for line in out.splitlines():
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+?): (.+?) \[-Rpass", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2).strip()
        if k == "Function Name":
            cur = {"name": v}
        else:
            cur[k] = v
        if k.startswith("LDS Size"):
            name = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(.*", "", name)
            print("%-60s vgpr %-4s agpr %-3s sgpr %-4s scratch %-5s occ %-3s lds %s" % (
                name[:60], cur.get("VGPRs"), cur.get("AGPRs"), cur.get("TotalSGPRs"),
                cur.get("ScratchSize [bytes/lane]"), cur.get("Occupancy [waves/SIMD]"), cur.get("LDS Size [bytes/block]")))
