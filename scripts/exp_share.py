"""Per-launch times of one launch chain of the headline scene: B frames, the tile share of one rank of N (timing
mode: every launch on one stream, bracketed by events)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracer_amd import Device, scenes, _abi
from raytracer_amd import dist as rd
from raytracer_amd._abi import Stats
KIND = {_abi.K_TRACE: "trace", _abi.K_SHADE: "shadow", _abi.K_COMBINE: "combine", _abi.K_SPAWN: "spawn", _abi.K_SCAN: "scan"}
w, h = 1920, 1080
world, cam = scenes.north_star(w, h)
dev = Device.get(0, 64)
scene = dev.upload(world)
B = int(os.environ.get("B", 1))
cds = (_abi.CameraDesc * B)(*[cam.to_desc() for _ in range(B)])
for n in [int(x) for x in os.environ.get("RANKS", "1,8").split(",")]:
    out = torch.empty(B * rd.tiles_per_rank(w, h, n) * 1024 * 3, dtype=torch.float32, device="cuda")
    st = Stats()
    dev.set_timing(True)
    acc = None
    for i in range(6):
        _abi.check(dev.lib.rt355_render_tiles_batch_begin(dev.handle, scene.handle, cds, B, 5, 0, n, C.c_void_p(out.data_ptr())))
        _abi.check(dev.lib.rt355_frame_end(dev.handle, C.byref(st)))
        if i:
            m = np.array(st.launch_ms[: st.n_launch]); acc = m if acc is None else acc + m
    dev.set_timing(False)
    acc /= 5
    print(f"== rank 0 of {n}, {B} frames per chain: rays by generation {[int(x) for x in st.radiance_rays][:6]}, sum of launches {acc.sum():.3f} ms = {acc.sum() / B:.3f} per frame")
    print("   " + "  ".join(f"{KIND.get(st.launch_kind[i], '?')}{st.launch_gen[i]}:{acc[i]*1e3:.0f}" for i in range(st.n_launch)))
