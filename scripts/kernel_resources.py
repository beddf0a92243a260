"""Per-kernel register, scratch, LDS and occupancy figures from hipcc's -Rpass-analysis=kernel-resource-usage
remarks.  usage: hipcc ... -Rpass-analysis=kernel-resource-usage ... 2> remarks.txt; python scripts/kernel_resources.py remarks.txt [filter]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0].strip().split()[0]
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if flt and flt not in dn:
        continue

    def g(k):
        m = re.search(re.escape(k) + r": (\d+)", b)
        return m.group(1) if m else "?"

    print("%-78s VGPR %3s AGPR %3s SGPR %3s scratch %4s occ %s LDS %s" % (
        dn[:78], g("VGPRs"), g("AGPRs"), g("SGPRs"), g("ScratchSize [bytes/lane]"), g("Occupancy [waves/SIMD]"), g("LDS Size [bytes/block]")))
