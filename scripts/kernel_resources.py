#!/usr/bin/env python3
"""Register use of every kernel of one instantiation of the launch chain, from the compiler's own report.
usage: scripts/kernel_resources.py [real=double] [mode=0] [csrc dir] [extra hipcc flags...]"""
import re
import subprocess
import sys

real = sys.argv[1] if len(sys.argv) > 1 else "double"
mode = sys.argv[2] if len(sys.argv) > 2 else "0"
csrc = sys.argv[3] if len(sys.argv) > 3 else "raytracer_amd/csrc"
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-fPIC", f"-DRT355_REAL={real}", f"-DRT355_MODE={mode}",
       "-c", "frame_inst.hip", "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[4:]
out = subprocess.run(cmd, cwd=csrc, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"^void rt355::|\(.*$", "", name)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([\w ]+?): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
print(f"{'kernel':52s} sgpr vgpr agpr s-spill v-spill scratch occ lds")
for r in rows:
    print(f"{r['name'][:52]:52s} {r.get('TotalSGPRs', 0):4d} {r.get('VGPRs', 0):4d} {r.get('AGPRs', 0):4d} {r.get('SGPRs Spill', 0):7d} {r.get('VGPRs Spill', 0):7d} "
          f"{r.get('ScratchSize', 0):7d} {r.get('Occupancy', 0):3d} {r.get('LDS Size', 0)}")
