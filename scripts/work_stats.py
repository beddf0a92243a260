"""Per-launch traversal work of one frame: box tests and primitive tests per ray (counting mode)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import Device, scenes
from raytracer_amd import dist as rd
from raytracer_amd._abi import Stats
world, cam = scenes.north_star(1920, 1080)
dev = Device.get(0, 64)
scene = dev.upload(world)
out = torch.empty(1920 * 1080 * 3, dtype=torch.float32, device="cuda")
st = Stats()
dev.set_counting(True)
rd.render_canvas(scene, cam, 5, out, st)
dev.set_counting(False)
names = {1: "trace", 2: "shadow", 3: "combine", 5: "spawn", 6: "scan"}
rays = [int(x) for x in st.radiance_rays]
hits = [int(x) for x in st.hits]
for i in range(st.n_launch):
    k, g = st.launch_kind[i], st.launch_gen[i]
    if k not in (1, 2, 5):
        continue
    n = rays[g] if k == 1 else hits[g]
    print(f"{names[k]:7s} gen {g}: rays {n:8d}  box tests/ray {st.launch_box_tests[i] / max(n, 1):7.1f}  prim tests/ray {st.launch_prim_tests[i] / max(n, 1):6.2f}  max box tests one lane {st.launch_max_box_tests[i]}")
