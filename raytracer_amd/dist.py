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


def untile_host(gathered, width, height, world_size, dtype=np.float32):
    """numpy counterpart of rt355_untile: `gathered` is (world_size, tiles_per_rank*1024*3)."""
    g = np.asarray(gathered, dtype=dtype).reshape(world_size, -1, TILE, TILE, 3)
    canvas = np.zeros((height * width, 3), dtype=dtype)
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


class FrameLoop:
    """One rank's side of a frame loop: render this rank's tiles, gather them to rank 0, un-tile there.

    `in_flight` frames are kept in flight on the GPU.  A frame is a chain of dependent launches, each ending on
    its slowest wave, so one frame alone leaves much of the device idle; with K > 1 every frame runs in a
    context of its own (one stream each, rt355_ctx_create_ex RT355_CTX_ONE_STREAM; the uploaded scene is
    shared) and `step` hands back the canvas of the frame begun K steps earlier.  `flush` completes what is
    still in flight.  With K = 1 a step renders and returns its own frame.

    Every slot owns its tile buffer.  With the nccl backend `dist.gather` only enqueues the send on a
    communication stream and returns, while the library renders on streams of its own: a rank other than 0 that
    rendered a later frame into the same buffer could overwrite tiles that are still being sent.  Each slot
    therefore remembers an event recorded behind its last send and does not render before that event has
    passed (in steady state it passed long ago).  Rank 0 synchronises with the gather before it un-tiles.

    `render(camera, out_tensor, stats)` writes this rank's tiles of the frame into the tensor it is given; the
    default calls the library (CUDA tensors); tests pass their own to run the same loop on CPU tensors over
    gloo.  `as_u8`: the tiles go through Color::to_u8 on the device before the gather (a quarter of the bytes)
    and the canvas is uint8, the body of a binary PPM.
    """

    def __init__(self, scene, device, width, height, rank, world_size, max_depth=5, group=None, tensor_device="cuda",
                 render=None, as_u8=False, in_flight=1):
        import torch
        self.scene, self.device, self.rank, self.world_size, self.group = scene, device, rank, world_size, group
        self.width, self.height, self.max_depth, self.as_u8 = width, height, max_depth, as_u8
        self.cuda = str(tensor_device).startswith("cuda")
        self.K = K = max(1, int(in_flight))
        self._custom = render
        n = tiles_per_rank(width, height, world_size) * TILE * TILE * 3
        out_dtype = torch.uint8 if as_u8 else torch.float32
        # with one rank and a float canvas the frame is written row-major in place: nothing to gather or un-tile
        self.direct = self.cuda and render is None and world_size == 1 and not as_u8
        if render is None and K > 1:
            from .host import Device
            self.ctxs = [Device(device.device_id, device.real_bits, flags=_abi.CTX_ONE_STREAM) for _ in range(K)]
            self._own_ctxs = True
        else:
            self.ctxs, self._own_ctxs = [device] * K, False
        self.tiles = [None if self.direct else torch.empty(n, dtype=torch.float32, device=tensor_device) for _ in range(K)]
        self.send = [torch.empty(n, dtype=out_dtype, device=tensor_device) for _ in range(K)] if as_u8 else self.tiles
        self.sent = [None] * K      # event behind the last send of each slot's buffer
        self.pending = [None] * K   # camera of the frame in flight in each slot
        self.gathered = torch.empty((world_size, n), dtype=out_dtype, device=tensor_device) if (rank == 0 and not self.direct) else None
        n_canvas = K if self.direct else 1
        self.canvases = [torch.empty((height, width, 3), dtype=out_dtype, device=tensor_device) for _ in range(n_canvas)] if rank == 0 else []
        self.canvas = self.canvases[0] if rank == 0 else None
        self.frame = 0

    def close(self):
        self.flush()
        if self._own_ctxs:
            for d in self.ctxs:
                d.close()
            self.ctxs, self._own_ctxs = [], False

    # -- one slot -------------------------------------------------------------------------------------------
    def _begin(self, k, camera):
        if self.sent[k] is not None:
            self.sent[k].synchronize()  # an earlier send read this slot's buffer
            self.sent[k] = None
        if self._custom is not None:
            self._custom(camera, self.tiles[k], None)
        else:
            ctx, cd = self.ctxs[k], camera.to_desc()
            if self.direct:
                _abi.check(ctx.lib.rt355_render_device_begin(ctx.handle, self.scene.handle, C.byref(cd), self.max_depth,
                                                             C.c_void_p(self.canvases[k].data_ptr())))
            else:
                _abi.check(ctx.lib.rt355_render_tiles_begin(ctx.handle, self.scene.handle, C.byref(cd), self.max_depth, self.rank, self.world_size,
                                                            C.c_void_p(self.tiles[k].data_ptr())))
        self.pending[k] = camera

    def _finish(self, k, stats):
        import torch
        import torch.distributed as dist
        camera, ctx = self.pending[k], self.ctxs[k]
        self.pending[k] = None
        if self._custom is None:
            _abi.check(ctx.lib.rt355_frame_end(ctx.handle, C.byref(stats) if stats is not None else None))  # the tiles are complete
        if self.direct:
            self.canvas = self.canvases[k]
            return self.canvas
        if self.as_u8:
            if self.cuda:
                _abi.check(ctx.lib.rt355_to_u8_device(ctx.handle, C.c_void_p(self.tiles[k].data_ptr()), self.tiles[k].numel(),
                                                      C.c_void_p(self.send[k].data_ptr())))
            else:
                from .host import Color
                self.send[k].copy_(torch.from_numpy(Color.to_u8(self.tiles[k].numpy())))
        if self.world_size > 1:
            dist.gather(self.send[k], list(self.gathered.unbind(0)) if self.rank == 0 else None, dst=0, group=self.group)
            if self.cuda and self.rank != 0:
                self.sent[k] = torch.cuda.Event()
                self.sent[k].record()  # torch's current stream, which the nccl work was ordered on
        elif self.rank == 0:
            self.gathered[0].copy_(self.send[k])
        if self.rank != 0:
            return None
        if self.cuda:
            fn = ctx.lib.rt355_untile_u8 if self.as_u8 else ctx.lib.rt355_untile
            torch.cuda.current_stream().synchronize()  # the gather has landed before the library's stream reads it
            cd = camera.to_desc()
            _abi.check(fn(ctx.handle, C.byref(cd), self.world_size, C.c_void_p(self.gathered.data_ptr()), C.c_void_p(self.canvas.data_ptr())))
        else:
            g = self.gathered.numpy()
            self.canvas.copy_(torch.from_numpy(untile_host(g, self.width, self.height, self.world_size, dtype=g.dtype)))
        return self.canvas

    # -- the loop -------------------------------------------------------------------------------------------
    def step(self, camera, stats=None):
        """Begins the frame of `camera`.  Returns, on rank 0, the canvas of the frame that completed in this step
        (valid until the next step): this frame with in_flight = 1, the frame begun in_flight steps ago otherwise
        (None while the pipeline fills); None on other ranks."""
        k = self.frame % self.K
        self.frame += 1
        done = None
        if self.K == 1:
            self._begin(k, camera)
            return self._finish(k, stats)
        if self.pending[k] is not None:
            done = self._finish(k, stats)
        self._begin(k, camera)
        return done

    def flush(self, stats=None):
        """Completes the frames still in flight, oldest first; returns their canvases (copies) on rank 0."""
        out = []
        for j in range(self.K):
            k = (self.frame + j) % self.K
            if self.pending[k] is not None:
                c = self._finish(k, stats)
                if c is not None:
                    out.append(c.clone())
        return out


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
