"""Frame time of rt355_render (canvas copied to host memory) next to rt355_render_tiles (canvas stays in
HBM): what the host-buffer boundary costs per frame."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import Device, scenes
from raytracer_amd import dist as rd
world, cam = scenes.north_star(1920, 1080)
dev = Device.get(0, 64)
scene = dev.upload(world)
out = torch.empty(rd.tiles_per_rank(1920, 1080, 1) * 1024 * 3, dtype=torch.float32, device="cuda")
for name, fn in (("render_tiles (HBM)", lambda: rd.render_tiles(scene, cam, 5, 0, 1, out)), ("render (host canvas)", lambda: scene.render(cam, 5))):
    for _ in range(3):
        fn()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    print(f"{name}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per frame")
