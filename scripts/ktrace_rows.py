import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
last = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for r in rows[-last:]:
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void rt355::", "")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0
    print(f"{name:40s} dur {(e - s) / 1e3:8.1f} us  gap {gap:7.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?'))} scratch {r.get('Scratch_Size', r.get('Private_Segment_Size', '?'))} lds {r.get('LDS_Block_Size', r.get('Group_Segment_Size', '?'))}")
    prev_end = e
