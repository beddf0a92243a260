"""Latency of ONE frame of the headline scene rendered as S tile shares at once (S contexts, each share a launch chain of
its own on its own stream, tiles interleaved like the ranks of a multi-GPU split), against the frame rendered by one
call.  Both end with the row-major canvas in device memory."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracer_amd import Device, scenes, _abi
from raytracer_amd import dist as rd

w, h = 1920, 1080
world, cam = scenes.north_star(w, h)
dev = Device.get(0, 64)
scene = dev.upload(world)
cd = cam.to_desc()
canvas = torch.empty(h * w * 3, dtype=torch.float32, device="cuda")
ref = torch.empty(h * w * 3, dtype=torch.float32, device="cuda")


def timed(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


one = lambda: rd.render_canvas(scene, cam, 5, ref)
print(f"one call: {timed(one):.4f} ms")
for flags, tag in ((_abi.CTX_ONE_STREAM, "one-stream contexts"), (0, "two-stream contexts")):
    for S in (1, 2, 3, 4):
        ctxs = [Device(0, 64, flags=flags) for _ in range(S)]
        tpr = rd.tiles_per_rank(w, h, S)
        tiles = torch.empty((S, tpr * 1024 * 3), dtype=torch.float32, device="cuda")

        def split():
            for r, c in enumerate(ctxs):
                _abi.check(dev.lib.rt355_render_tiles_begin(c.handle, scene.handle, C.byref(cd), 5, r, S, C.c_void_p(tiles[r].data_ptr())))
            for c in ctxs:
                _abi.check(dev.lib.rt355_frame_end(c.handle, None))
            rd.untile(dev, cam, S, tiles, canvas)

        ms = timed(split)
        one()
        torch.cuda.synchronize()
        same = bool(torch.equal(canvas, ref))
        print(f"{tag}, {S} shares: {ms:.4f} ms  identical to one call: {same}")
        for c in ctxs:
            c.close()
