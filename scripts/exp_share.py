"""Per-launch times of one frame of the headline scene for the tile share of one rank of N (timing mode: every
launch on one stream, bracketed by events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracer_amd import Device, scenes
from raytracer_amd import dist as rd
from raytracer_amd._abi import Stats
KIND = {0: "reset", 1: "trace", 2: "scan", 3: "spawn", 4: "shadow", 5: "combine", 6: "untile", 7: "trace+shade"}
w, h = 1920, 1080
world, cam = scenes.north_star(w, h)
dev = Device.get(0, 64)
scene = dev.upload(world)
for n in [int(x) for x in os.environ.get("RANKS", "1,8").split(",")]:
    out = torch.empty(rd.tiles_per_rank(w, h, n) * 1024 * 3, dtype=torch.float32, device="cuda")
    st = Stats()
    dev.set_timing(True)
    acc = None
    for i in range(6):
        rd.render_tiles(scene, cam, 5, 0, n, out, st)
        if i:
            m = np.array(st.launch_ms[: st.n_launch]); acc = m if acc is None else acc + m
    dev.set_timing(False)
    acc /= 5
    print(f"== rank 0 of {n}: rays by generation {[int(x) for x in st.radiance_rays][:6]}, sum of launches {acc.sum():.3f} ms")
    print("   " + "  ".join(f"{KIND.get(st.launch_kind[i], '?')}{st.launch_gen[i]}:{acc[i]*1e3:.0f}" for i in range(st.n_launch)))
