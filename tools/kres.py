#!/usr/bin/env python3
"""Print a per-kernel resource table (VGPR/SGPR/LDS/scratch/occupancy) for a HIP source.
usage: tools/kres.py file.hip [extra hipcc flags]"""
import re, subprocess, sys
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "--offload-arch=gfx950", "-std=c++17", "-I", "include", "-I", "../../include",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], {}
for line in out.splitlines():
    m = re.search(r"remark:\s+([A-Za-z \[\]/]+?): (.+?) \[-Rpass", line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()[:70]}
        rows.append(cur)
    else: cur[k] = v
print(f"{'kernel':70s} {'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'LDS':>7} {'scr':>5} {'occ':>4}")
for r in rows:
    print(f"{r['name']:70s} {r.get('VGPRs','?'):>5} {r.get('AGPRs','?'):>5} {r.get('TotalSGPRs','?'):>5} {r.get('LDS Size [bytes/block]','?'):>7} {r.get('ScratchSize [bytes/lane]','?'):>5} {r.get('Occupancy [waves/SIMD]','?'):>4}")
