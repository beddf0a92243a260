"""Renders a few frames of one scene through the C ABI; the program rocprofv3 wraps for PMC passes."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402

from raytracer_amd import Device, scenes  # noqa: E402
from raytracer_amd import dist as rd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="north_star")
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--frames", type=int, default=3, help="launch chains to run")
ap.add_argument("--batch", type=int, default=16, help="frames per launch chain (the bench default at one GPU)")
ap.add_argument("--real", type=int, default=64)
ap.add_argument("--single", action="store_true", help="one frame per call, begun and waited for (rt355_render_tiles): the plan of a frame rendered alone")
ap.add_argument("--keep", default="", help="diagnostic: keep only these shapes of the scene: floor, sphere, mesh (comma list); 'matte' also makes every material non-reflective and opaque")
a = ap.parse_args()
world, cam = scenes.CONFIGS[a.scene](a.width, a.height)
if a.keep:
    keep = set(a.keep.split(","))
    from raytracer_amd import _abi
    kinds = {"floor": (_abi.PLANE,), "sphere": (_abi.SPHERE,), "mesh": (_abi.TRIANGLE, _abi.SMOOTH_TRIANGLE)}
    wanted = {k for name in keep if name in kinds for k in kinds[name]}
    world.shapes = [s for s in world.shapes if s.kind in wanted]
    if "matte" in keep:
        for s in world.shapes:
            s.material.reflective = 0.0
            s.material.transparency = 0.0
    print("shapes kept:", len(world.shapes))
dev = Device.get(0, a.real)
scene = dev.upload(world)
out = torch.empty(a.batch * rd.tiles_per_rank(a.width, a.height, 1) * 1024 * 3, dtype=torch.float32, device="cuda")
for _ in range(a.frames):
    if a.single:
        rd.render_tiles(scene, cam, 5, 0, 1, out)
    else:
        rd.render_tiles_chain(scene, [cam] * a.batch, 5, 0, 1, out)  # a chain of a frame loop
torch.cuda.synchronize()
print("frames done")
