"""Tile split of one frame across ranks and the canvas gather (one process per GPU).

Pixels are independent in the reference (`Camera::render` has no cross-pixel state,
src/camera.rs:76-82), so the frame shards naturally: the canvas is cut into 32x32 tiles, tile t
belongs to rank t % world_size (round-robin, so expensive image regions spread over all ranks), the
scene is replicated, and the only exchange is ONE gather of the rank tile buffers to rank 0
(`torch.distributed.gather`, RCCL over xGMI with the nccl backend), followed by an un-tiling kernel on
rank 0.  No collective runs during rendering.
"""
import ctypes as C

import numpy as np

from . import _abi

TILE = 32


def tiles_per_rank(width, height, world_size):
    tx, ty = (width + TILE - 1) // TILE, (height + TILE - 1) // TILE
    return (tx * ty + world_size - 1) // world_size


def tile_pixels(width, height, rank, world_size):
    """Row-major pixel index (y*width+x, or -1 for padding) of every slot of this rank's tile buffer:
    shape (tiles_per_rank, 32, 32).  Host-side statement of the layout rt355_render_tiles writes."""
    tx, ty = (width + TILE - 1) // TILE, (height + TILE - 1) // TILE
    tpr = tiles_per_rank(width, height, world_size)
    out = np.full((tpr, TILE, TILE), -1, dtype=np.int64)
    yy, xx = np.mgrid[0:TILE, 0:TILE]
    for tl in range(tpr):
        t = tl * world_size + rank
        if t >= tx * ty:
            continue
        x = (t % tx) * TILE + xx
        y = (t // tx) * TILE + yy
        ok = (x < width) & (y < height)
        out[tl][ok] = (y * width + x)[ok]
    return out


def untile_host(gathered, width, height, world_size):
    """numpy counterpart of rt355_untile: `gathered` is (world_size, tiles_per_rank*1024*3)."""
    g = np.asarray(gathered, dtype=np.float32).reshape(world_size, -1, TILE, TILE, 3)
    canvas = np.zeros((height * width, 3), dtype=np.float32)
    for r in range(world_size):
        idx = tile_pixels(width, height, r, world_size)
        ok = idx >= 0
        canvas[idx[ok]] = g[r][ok]
    return canvas.reshape(height, width, 3)


def render_tiles(scene, camera, max_depth, rank, world_size, out_tensor, stats=None):
    """This rank's tiles into `out_tensor` (a CUDA float32 tensor of tiles_per_rank*1024*3 elements)."""
    cd = camera.to_desc()
    need = tiles_per_rank(int(camera.horizontal_size), int(camera.vertical_size), world_size) * TILE * TILE * 3
    assert out_tensor.is_cuda and out_tensor.is_contiguous() and out_tensor.numel() == need
    _abi.check(scene.lib.rt355_render_tiles(scene.device.handle, scene.handle, C.byref(cd), max_depth, rank, world_size,
                                            C.c_void_p(out_tensor.data_ptr()), C.byref(stats) if stats is not None else None))


def render_canvas(scene, camera, max_depth, out_tensor, stats=None):
    """The whole frame, row-major, into `out_tensor` (a CUDA float32 tensor of height*width*3 elements):
    what a single rank does, where there is nothing to gather or un-tile."""
    cd = camera.to_desc()
    need = int(camera.horizontal_size) * int(camera.vertical_size) * 3
    assert out_tensor.is_cuda and out_tensor.is_contiguous() and out_tensor.numel() == need
    _abi.check(scene.lib.rt355_render_device(scene.device.handle, scene.handle, C.byref(cd), max_depth,
                                             C.c_void_p(out_tensor.data_ptr()), C.byref(stats) if stats is not None else None))


def untile(device, camera, world_size, gathered, out_tensor):
    """Tile buffers of all ranks (CUDA tensor, rank-major) -> row-major canvas, on rank 0.  The library
    works on its own stream, so whatever produced `gathered` on torch's streams (the gather) is waited
    for here first."""
    import torch
    assert gathered.is_cuda and out_tensor.is_cuda and gathered.is_contiguous() and out_tensor.is_contiguous()
    torch.cuda.current_stream(gathered.device).synchronize()
    cd = camera.to_desc()
    _abi.check(device.lib.rt355_untile(device.handle, C.byref(cd), world_size, C.c_void_p(gathered.data_ptr()),
                                       C.c_void_p(out_tensor.data_ptr())))


def gather_canvas(local_tiles, width, height, rank, world_size, group=None, device=None, camera=None):
    """Gather every rank's tile buffer to rank 0 and un-tile.  CUDA tensors go through
    torch.distributed.gather (RCCL) + the rt355_untile kernel; CPU tensors (gloo, tests) through the
    numpy restatement.  Returns the (height, width, 3) canvas on rank 0, None elsewhere."""
    import torch
    import torch.distributed as dist

    if world_size == 1:
        gathered = local_tiles.reshape(1, -1)
    else:
        bufs = [torch.empty_like(local_tiles) for _ in range(world_size)] if rank == 0 else None
        dist.gather(local_tiles, bufs, dst=0, group=group)
        if rank != 0:
            return None
        gathered = torch.stack(bufs, 0)
    if gathered.is_cuda:
        out = torch.empty((height, width, 3), dtype=torch.float32, device=gathered.device)
        untile(device, camera, world_size, gathered.contiguous(), out)
        return out
    return torch.from_numpy(untile_host(gathered.numpy(), width, height, world_size))


def canvas_to_u8(device, canvas):
    """Color::to_u8 of a canvas that lives in HBM (CUDA float32 tensor) -> CUDA uint8 tensor of the same shape:
    the body of a binary PPM, produced where the canvas is."""
    import torch
    assert canvas.is_cuda and canvas.is_contiguous() and canvas.dtype == torch.float32
    torch.cuda.current_stream(canvas.device).synchronize()
    out = torch.empty(canvas.shape, dtype=torch.uint8, device=canvas.device)
    _abi.check(device.lib.rt355_to_u8_device(device.handle, C.c_void_p(canvas.data_ptr()), canvas.numel(), C.c_void_p(out.data_ptr())))
    return out
