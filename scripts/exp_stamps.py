"""Diagnostic: per-wave traversal time of the trace launches of one frame, split into node steps, triangle leaves
and other leaves (needs the RT355_EXP_STAMPS build via RT355_LIB)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracer_amd import Device, scenes, _abi
from raytracer_amd import dist as rd
from raytracer_amd._abi import Stats
w, h = int(os.environ.get("W", 1920)), int(os.environ.get("H", 1080))
world, cam = scenes.CONFIGS[os.environ.get("SCENE", "north_star")](w, h)
dev = Device.get(0, 64)
dev.set_timing(True)   # all launches on one stream
scene = dev.upload(world)
out = torch.empty(h * w * 3, dtype=torch.float32, device="cuda")
lib = _abi.lib()
rec = np.dtype([("t0", "<u8"), ("t1", "<u8"), ("cn", "<u8"), ("ct", "<u8"), ("cp", "<u8"), ("ctot", "<u8"), ("trips", "<u4"), ("lanes", "<u4"), ("nn", "<u4"), ("nl", "<u4")])
PER = 1 << 15
buf = np.zeros(48 * PER, dtype=rec)
lib.rt355_dbg_read.restype = C.c_int
lib.rt355_dbg_read.argtypes = [C.c_void_p, C.c_size_t]
st = Stats()
for _ in range(3):
    rd.render_canvas(scene, cam, 5, out, st)
assert lib.rt355_dbg_read(C.c_void_p(buf.ctypes.data), buf.nbytes) == 0
rd.render_canvas(scene, cam, 5, out, st)
assert lib.rt355_dbg_read(C.c_void_p(buf.ctypes.data), buf.nbytes) == 0
for li in range(st.n_launch):
    if st.launch_kind[li] != 1:
        continue
    r = buf[li * PER:(li + 1) * PER]
    r = r[r["trips"] > 0]
    if len(r) == 0:
        continue
    t_origin = r["t0"].min()
    dur = (r["t1"] - r["t0"]) / 100.0
    end = (r["t1"] - t_origin) / 100.0
    k = int(np.argmax(end))
    tot = r["ctot"].astype(np.float64)
    print(f"trace gen {st.launch_gen[li]} ({st.launch_ms[li]*1e3:.0f} us): waves {len(r)} span {end.max():.1f} us; wave dur p50 {np.percentile(dur,50):.1f} p99 {np.percentile(dur,99):.1f} max {dur.max():.1f} us; "
          f"all waves: cycles node {r['cn'].sum()/tot.sum():.2f} tri {r['ct'].sum()/tot.sum():.2f} prim {r['cp'].sum()/tot.sum():.2f} of traversal; cycles per node step {r['cn'].sum()/max(1,r['nn'].sum()):.0f}, per leaf step {(r['ct'].sum()+r['cp'].sum())/max(1,r['nl'].sum()):.0f}")
    top = np.argsort(-end)[:3]
    for j in top:
        q = r[j]
        print(f"    late wave: start {(q['t0']-t_origin)/100.0:.1f} end {(q['t1']-t_origin)/100.0:.1f} us, trips {q['trips']} (node {q['nn']}, leaf {q['nl']}), lanes {q['lanes']}, cycles total {q['ctot']} node {q['cn']} tri {q['ct']} prim {q['cp']} -> {q['ctot']/max(1,q['trips']):.0f} cycles/trip")
