"""Average duration of every launch of a chain, by its place in the chain, from a rocprofv3 kernel trace of bench.py:
   python scripts/ktrace_by_generation.py <..._kernel_trace.csv>
One symbol serves several generations (k_trace<double, false, true, false> traces generations 0..4), so the
per-symbol averages of --stats mix launches of very different sizes; this table keeps them apart.  Chains are cut at
k_reset_ctl; only chains with the most common launch sequence are averaged (the timed loop's), the instrumented
COUNT kernels and the single-frame chains of the bench's other measurements fall out."""
import csv, re, sys
from collections import Counter, defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
by_q = defaultdict(list)
for r in rows:
    by_q[r.get("Queue_Id", "0")].append(r)
chains = []
for q, rs in by_q.items():
    cur = None
    for r in rs:
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void rt355::", "").replace("rt355::", "")
        if not name.startswith("k_"):
            continue
        if name == "k_reset_ctl":
            if cur:
                chains.append(cur)
            cur = []
            continue
        if cur is not None:
            cur.append((name, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)))
    if cur:
        chains.append(cur)
sig = Counter(tuple((n, g) for n, _, g in c) for c in chains)
best, n_best = sig.most_common(1)[0]
sel = [c for c in chains if tuple((n, g) for n, _, g in c) == best]
print(f"# {len(chains)} chains in the trace, {n_best} with the most common launch sequence (same kernels, same grids); average duration per launch of those")
seen = Counter()
for i, (name, grid) in enumerate(best):
    base = name.split("<")[0]
    gen = seen[base]
    seen[base] += 1
    d = [c[i][1] for c in sel]
    print(f"{i:3d} {name:38s} {base} launch #{gen} of the chain   avg {sum(d) / len(d):9.1f} us   min {min(d):9.1f}   max {max(d):9.1f}   grid {grid}")
