"""Experiment: frame time of the north-star scene against resolution (launch-chain floor vs work)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import Device, scenes
from raytracer_amd import dist as rd
from raytracer_amd._abi import Stats
dev = Device.get(0, 64)
scene = None
for (w, h) in [(64, 36), (240, 135), (480, 270), (960, 540), (1920, 1080), (3840, 2160)]:
    world, cam = scenes.north_star(w, h)
    if scene is None:
        scene = dev.upload(world)
    out = torch.empty(h * w * 3, dtype=torch.float32, device="cuda")
    st = Stats()
    for _ in range(5):
        rd.render_canvas(scene, cam, 5, out, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 30
    for _ in range(n):
        rd.render_canvas(scene, cam, 5, out, st)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    dev.set_timing(True)
    rd.render_canvas(scene, cam, 5, out, st)
    dev.set_timing(False)
    ls = " ".join("%d:%.0f" % (st.launch_kind[i], st.launch_ms[i] * 1e3) for i in range(st.n_launch))
    print(f"{w}x{h}: {ms:.3f} ms/frame, rays {int(st.total_radiance + st.total_shadow)}, serial launches us: {ls}", flush=True)
