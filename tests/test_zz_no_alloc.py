"""rt355_render_device_noalloc: the entry point that allocates nothing (its ray pool is a fixed workspace inside
the library) against rt355_render_device, through the C ABI.  Two calls in a row, the second with a smaller
frame: the workspace keeps the first frame's contents, the second result must not depend on them.  Bar: the
canvases are bit-identical and the ray statistics equal (same kernels, same arithmetic, another place for the
pool)."""
import ctypes as C

import numpy as np
import pytest

from raytracer_amd import Device, _abi, scenes
from raytracer_amd._abi import Rt355Error, Stats

pytestmark = pytest.mark.gpu


def render_both(dev, scene, cam):
    import torch
    w, h = int(cam.horizontal_size), int(cam.vertical_size)
    cd = cam.to_desc()
    out = []
    for fn in (dev.lib.rt355_render_device, dev.lib.rt355_render_device_noalloc):
        canvas = torch.full((h, w, 3), -7.0, dtype=torch.float32, device="cuda")
        st = Stats()
        _abi.check(fn(dev.handle, scene.handle, C.byref(cd), 5, C.c_void_p(canvas.data_ptr()), C.byref(st)))
        out.append((canvas.cpu().numpy(), [int(x) for x in st.radiance_rays], [int(x) for x in st.hits]))
    return out


@pytest.mark.parametrize("real_bits", [64, 32])
def test_noalloc_equals_render_device(real_bits):
    dev = Device.get(0, real_bits)
    world, _ = scenes.glass(8, 8)
    scene = dev.upload(world)
    try:
        for size in ((160, 120), (96, 64), (33, 31)):  # a larger frame first, then smaller ones over its leftovers
            _, cam = scenes.glass(*size)
            (a, ra, ha), (b, rb, hb) = render_both(dev, scene, cam)
            assert np.array_equal(a, b, equal_nan=True), size
            assert ra == rb and ha == hb and sum(ra) > size[0] * size[1]
            assert not (b == -7.0).any()
    finally:
        scene.close()


def test_noalloc_rejects_frames_beyond_the_workspace():
    import torch
    dev = Device.get(0, 64)
    world, cam = scenes.glass(1920, 1080)  # 2.1 M primary rays x 8 x 122 bytes: far beyond 64 MiB
    scene = dev.upload(world)
    canvas = torch.zeros((1080, 1920, 3), dtype=torch.float32, device="cuda")
    cd = cam.to_desc()
    try:
        with pytest.raises(Rt355Error, match="fixed workspace"):
            _abi.check(dev.lib.rt355_render_device_noalloc(dev.handle, scene.handle, C.byref(cd), 5, C.c_void_p(canvas.data_ptr()), None))
        assert not canvas.any()  # refused before any launch
        # the same context still renders through the ordinary entry point, and small frames through this one
        _, small = scenes.glass(64, 48)
        (a, _, _), (b, _, _) = render_both(dev, scene, small)
        assert np.array_equal(a, b)
    finally:
        scene.close()


def test_noalloc_allocates_nothing_on_a_context_that_never_rendered():
    """A context that has rendered nothing owns no ray pool.  The no-allocation entry point alone must leave the
    device's free memory where it was -- over its first frame and over a second, larger one -- which a silent
    fall-back to the growable pool could not.  The canvases are then held to rt355_render_device of ANOTHER context."""
    import torch
    fresh = Device(0, 64)          # its own context: no pool yet
    other = Device.get(0, 64)
    try:
        world, _ = scenes.glass(8, 8)
        scene = fresh.upload(world)
        sizes = ((64, 48), (160, 120))
        cams = [scenes.glass(*sz)[1] for sz in sizes]
        canvases = [torch.full((sz[1], sz[0], 3), -7.0, dtype=torch.float32, device="cuda") for sz in sizes]
        fresh.lib.rt355_render_device_noalloc(fresh.handle, scene.handle, C.byref(cams[0].to_desc()), 0, C.c_void_p(canvases[0].data_ptr()), None)  # warm-up: the runtime sizes its per-kernel private memory on first launches
        _abi.check(fresh.lib.rt355_render_device_noalloc(fresh.handle, scene.handle, C.byref(cams[0].to_desc()), 5, C.c_void_p(canvases[0].data_ptr()), None))
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info()[0]
        for cam, canvas in zip(cams, canvases):
            cd = cam.to_desc()
            _abi.check(fresh.lib.rt355_render_device_noalloc(fresh.handle, scene.handle, C.byref(cd), 5, C.c_void_p(canvas.data_ptr()), None))
            torch.cuda.synchronize()
            assert torch.cuda.mem_get_info()[0] == free_before, "the no-allocation entry point changed the device's free memory"
        ref_scene = other.upload(world)
        try:
            for cam, canvas, sz in zip(cams, canvases, sizes):
                want = torch.empty((sz[1], sz[0], 3), dtype=torch.float32, device="cuda")
                _abi.check(other.lib.rt355_render_device(other.handle, ref_scene.handle, C.byref(cam.to_desc()), 5, C.c_void_p(want.data_ptr()), None))
                assert np.array_equal(canvas.cpu().numpy(), want.cpu().numpy())
        finally:
            ref_scene.close()
        scene.close()
    finally:
        fresh.close()
