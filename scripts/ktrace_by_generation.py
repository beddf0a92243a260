"""Average duration of every launch of a chain, by its place in the chain, from a rocprofv3 kernel trace of bench.py:
   python scripts/ktrace_by_generation.py <..._kernel_trace.csv>
One symbol serves several generations (k_trace<double, false, true, false> traces generations 0..4), so the
per-symbol averages of --stats mix launches of very different sizes; this table keeps them apart.  Chains are cut at
k_reset_ctl; the two most common launch sequences are averaged (the timed loop's chains and the single-frame chains
of the bench's other measurements); the instrumented counting kernels fall out."""
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
# the two most common sequences: the timed loop's chains (several frames each) and the single-frame chains of the
# bench's other measurements (one frame rendered alone); told apart by the grid of their first launch
for best, n_best in sig.most_common(2):
    sel = [c for c in chains if tuple((n, g) for n, _, g in c) == best]
    total = sum(sum(c[i][1] for c in sel) / len(sel) for i in range(len(best)))
    print(f"# {len(chains)} chains in the trace, {n_best} with this launch sequence (same kernels, same grids): average duration per launch; "
          f"sum {total:.1f} us; first grid {best[0][1] if best else 0}")
    seen = Counter()
    for i, (name, grid) in enumerate(best):
        base = name.split("<")[0]
        gen = seen[base]
        seen[base] += 1
        d = [c[i][1] for c in sel]
        print(f"{i:3d} {name:38s} {base} launch #{gen} of the chain   avg {sum(d) / len(d):9.1f} us   min {min(d):9.1f}   max {max(d):9.1f}   grid {grid}")
    print()
