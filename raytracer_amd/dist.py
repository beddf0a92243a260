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


def render_tiles_chain(scene, cameras, max_depth, rank, world_size, out_tensor, stats=None):
    """One launch chain the way a frame loop runs it (rt355_render_tiles_batch_begin + rt355_frame_end: begun
    without waiting, one or several frames of one size), for measuring that plan one chain at a time.  `cameras`: a
    camera or a list of them; `out_tensor` holds the tiles of all the frames, frame after frame."""
    cams = list(cameras) if isinstance(cameras, (list, tuple)) else [cameras]
    cds = (_abi.CameraDesc * len(cams))(*[c.to_desc() for c in cams])
    need = len(cams) * tiles_per_rank(int(cams[0].horizontal_size), int(cams[0].vertical_size), world_size) * TILE * TILE * 3
    assert out_tensor.is_cuda and out_tensor.is_contiguous() and out_tensor.numel() == need
    _abi.check(scene.lib.rt355_render_tiles_batch_begin(scene.device.handle, scene.handle, cds, len(cams), max_depth, rank, world_size,
                                                        C.c_void_p(out_tensor.data_ptr())))
    _abi.check(scene.lib.rt355_frame_end(scene.device.handle, C.byref(stats) if stats is not None else None))


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

    A frame is a chain of dependent launches, each ending on its slowest wave, so one frame alone -- and all the
    more the few tiles one rank of many renders of it -- leaves much of the device idle.  Two things fill it:

    `batch` frames share ONE launch chain (rt355_render_tiles_batch_begin): `step` collects cameras and launches
    when it holds `batch` of them; the rays of all the frames are traced together, their tiles are gathered in
    one collective and un-tiled in one launch.

    `in_flight` chains are kept in flight, each in a context of its own (one stream each, rt355_ctx_create_ex
    RT355_CTX_ONE_STREAM; the uploaded scene is shared).

    `step` hands back finished canvases in the order the frames were begun, one per call, or None while the
    pipeline fills; `flush` completes what is still collected or in flight.  With batch = in_flight = 1 a step
    renders and returns its own frame.

    Every slot owns its tile buffer.  With the nccl backend `dist.gather` only enqueues the send on a
    communication stream and returns, while the library renders on streams of its own: a rank other than 0 that
    rendered later frames into the same buffer could overwrite tiles that are still being sent.  Each slot
    therefore remembers an event recorded behind its last send and does not render before that event has
    passed (in steady state it passed long ago).  On rank 0 the un-tiling is ordered behind the gather, and the
    caller's stream behind the un-tiling, by events: the canvases a step hands back are valid for work queued on
    torch's current stream (or after torch.cuda.synchronize()), and the host waits only in rt355_frame_end.

    `render(camera, out_tensor, stats)` writes this rank's tiles of one frame into the tensor it is given; the
    default calls the library (CUDA tensors); tests pass their own to run the same loop on CPU tensors over
    gloo.  `as_u8`: the tiles go through Color::to_u8 on the device before the gather (a quarter of the bytes)
    and the canvas is uint8, the body of a binary PPM.
    """

    def __init__(self, scene, device, width, height, rank, world_size, max_depth=5, group=None, tensor_device="cuda",
                 render=None, as_u8=False, in_flight=1, batch=1):
        import collections
        import torch
        self.scene, self.device, self.rank, self.world_size, self.group = scene, device, rank, world_size, group
        self.width, self.height, self.max_depth, self.as_u8 = width, height, max_depth, as_u8
        self.cuda = str(tensor_device).startswith("cuda")
        self.K = K = max(1, int(in_flight))
        self.B = B = max(1, min(int(batch), _abi.MAX_BATCH))
        self._custom = render
        self.n = n = tiles_per_rank(width, height, world_size) * TILE * TILE * 3  # floats of one frame's tiles on one rank
        self.out_dtype = out_dtype = torch.uint8 if as_u8 else torch.float32
        self.tensor_device = tensor_device
        # with one rank and a float canvas the frames are written row-major in place: nothing to gather or un-tile
        self.direct = self.cuda and render is None and world_size == 1 and not as_u8
        if render is None and (K > 1 or B > 1):  # chains meant to overlap or to carry several frames: one stream each
            from .host import Device
            self.ctxs = [Device(device.device_id, device.real_bits, flags=_abi.CTX_ONE_STREAM) for _ in range(K)]
            self._own_ctxs = True
        else:
            self.ctxs, self._own_ctxs = [device] * K, False
        self.tiles = [None if self.direct else torch.empty(B * n, dtype=torch.float32, device=tensor_device) for _ in range(K)]
        self.send = [torch.empty(B * n, dtype=out_dtype, device=tensor_device) for _ in range(K)] if as_u8 else self.tiles
        self.sent = [None] * K      # event behind the last send of each slot's buffer
        self.pending = [None] * K   # cameras of the chain in flight in each slot
        self.where = [0] * K        # direct mode: which canvas set the slot's chain writes
        self.gathered = torch.empty((world_size, B * n), dtype=out_dtype, device=tensor_device) if (rank == 0 and not self.direct) else None
        # direct mode renders into the canvases themselves: one set more than chains in flight, so the set whose
        # canvases are being handed out is never the one the next chain writes
        n_sets = K + 1 if self.direct else 1
        self.canvases = [torch.empty((B, height, width, 3), dtype=out_dtype, device=tensor_device) for _ in range(n_sets)] if rank == 0 else []
        self.canvas = self.canvases[0][0] if rank == 0 else None
        self.cams = []              # cameras collected for the next chain
        self.ready = collections.deque()
        self.launches = 0
        self.frame = 0
        self.stats_frames = 0       # how many frames the statistics of the last finished chain cover
        self.rays_traced = 0        # radiance + shadow rays of all finished frames (library renders only), tile padding excluded
        self.frames_done = 0
        self._active = None         # pixels of a frame this rank owns

    def close(self):
        self.flush()
        if self._own_ctxs:
            for d in self.ctxs:
                d.close()
            self.ctxs, self._own_ctxs = [], False

    # -- one slot -------------------------------------------------------------------------------------------
    def _begin(self, k, cameras):
        nf = len(cameras)
        if self.sent[k] is not None:
            self.sent[k].synchronize()  # an earlier send read this slot's buffer
            self.sent[k] = None
        if self._custom is not None:
            for f, camera in enumerate(cameras):
                self._custom(camera, self.tiles[k][f * self.n:(f + 1) * self.n], None)
        else:
            ctx = self.ctxs[k]
            cds = (_abi.CameraDesc * nf)(*[c.to_desc() for c in cameras])
            if self.direct:
                self.where[k] = self.launches % len(self.canvases)
                _abi.check(ctx.lib.rt355_render_device_batch_begin(ctx.handle, self.scene.handle, cds, nf, self.max_depth,
                                                                   C.c_void_p(self.canvases[self.where[k]].data_ptr())))
            else:
                _abi.check(ctx.lib.rt355_render_tiles_batch_begin(ctx.handle, self.scene.handle, cds, nf, self.max_depth, self.rank, self.world_size,
                                                                  C.c_void_p(self.tiles[k].data_ptr())))
        self.launches += 1
        self.pending[k] = list(cameras)

    def _finish(self, k, stats):
        """Completes the chain of slot k; on rank 0 its canvases join `ready`."""
        import torch
        import torch.distributed as dist
        cameras, ctx = self.pending[k], self.ctxs[k]
        nf, n = len(cameras), self.n
        self.pending[k] = None
        if self._custom is None:
            if stats is None:
                stats = _abi.Stats()
            _abi.check(ctx.lib.rt355_frame_end(ctx.handle, C.byref(stats)))  # the tiles are complete
            if self._active is None:
                self._active = int((tile_pixels(self.width, self.height, self.rank, self.world_size) >= 0).sum())
            # generation 0 counts every slot of the rank's tiles; the slots beyond the image edge hold no ray
            self.rays_traced += int(stats.total_radiance + stats.total_shadow) - (int(stats.radiance_rays[0]) - nf * self._active)
        self.frames_done += nf
        self.stats_frames = nf
        if self.direct:
            self.ready.extend(self.canvases[self.where[k]][f] for f in range(nf))
            return
        tiles, send = self.tiles[k][: nf * n], self.send[k][: nf * n]
        # Between here and the canvases the host waits for nothing: the conversion and the un-tiling are only queued
        # on the library's stream, the collective on torch's, and the two are ordered by events
        # (rt355_stream_wait_ctx / rt355_ctx_wait_stream).
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream) if self.cuda else None
        if self.cuda:
            _abi.check(ctx.lib.rt355_ctx_set_async(ctx.handle, 1))
        try:
            self._gather_and_untile(k, ctx, cameras, tiles, send, stream)
        finally:
            if self.cuda:
                _abi.check(ctx.lib.rt355_ctx_set_async(ctx.handle, 0))

    def _gather_and_untile(self, k, ctx, cameras, tiles, send, stream):
        import torch
        import torch.distributed as dist
        nf, n = len(cameras), self.n
        if self.as_u8:
            if self.cuda:
                _abi.check(ctx.lib.rt355_to_u8_device(ctx.handle, C.c_void_p(tiles.data_ptr()), tiles.numel(), C.c_void_p(send.data_ptr())))
                _abi.check(ctx.lib.rt355_stream_wait_ctx(ctx.handle, stream))  # the send reads the converted tiles
            else:
                from .host import Color
                send.copy_(torch.from_numpy(Color.to_u8(tiles.numpy())))
        gathered = None
        if self.rank == 0:  # a chain shorter than `batch` (the last one) is gathered into a buffer of its own size
            gathered = self.gathered if nf == self.B else torch.empty((self.world_size, nf * n), dtype=self.out_dtype, device=self.tensor_device)
        if self.world_size > 1:
            dist.gather(send, list(gathered.unbind(0)) if self.rank == 0 else None, dst=0, group=self.group)
            if self.cuda and self.rank != 0:
                self.sent[k] = torch.cuda.Event()
                self.sent[k].record()  # torch's current stream, which the nccl work was ordered on
        elif self.rank == 0:
            gathered[0].copy_(send)
        if self.rank != 0:
            return
        canvases = self.canvases[0]
        if self.cuda:
            fn = ctx.lib.rt355_untile_batch_u8 if self.as_u8 else ctx.lib.rt355_untile_batch
            # the library's stream waits for the gather (and for whatever the caller still has queued on the canvases)
            _abi.check(ctx.lib.rt355_ctx_wait_stream(ctx.handle, stream))
            cd = cameras[0].to_desc()
            _abi.check(fn(ctx.handle, C.byref(cd), self.world_size, nf, C.c_void_p(gathered.data_ptr()), C.c_void_p(canvases.data_ptr())))
            # what the caller queues on torch's stream from here on (its use of the canvases, the next gather into the
            # shared buffer) starts behind the un-tiling
            _abi.check(ctx.lib.rt355_stream_wait_ctx(ctx.handle, stream))
        else:
            g = gathered.numpy()
            for f in range(nf):
                canvases[f].copy_(torch.from_numpy(untile_host(g[:, f * n:(f + 1) * n], self.width, self.height, self.world_size, dtype=g.dtype)))
        self.ready.extend(canvases[f] for f in range(nf))

    def _launch(self, stats):
        k = self.launches % self.K
        if self.pending[k] is not None:
            self._finish(k, stats)
        cameras, self.cams = self.cams, []
        self._begin(k, cameras)
        if self.K == 1 and self.B == 1:
            self._finish(k, stats)

    # -- the loop -------------------------------------------------------------------------------------------
    def step(self, camera, stats=None):
        """Adds the frame of `camera`.  Returns, on rank 0, the next finished canvas in the order the frames were
        begun (valid until the next step), or None while the pipeline fills; always None on other ranks.  With
        batch = in_flight = 1 that is this very frame."""
        self.frame += 1
        self.cams.append(camera)
        if len(self.cams) == self.B:
            self._launch(stats)
        if self.ready:
            self.canvas = self.ready.popleft()
            return self.canvas
        return None

    def flush(self, stats=None):
        """Launches the cameras still collected and completes every chain in flight, oldest first; returns, on
        rank 0, the canvases not yet handed out (copies)."""
        out = []

        def drain():  # chains share the canvas set they are un-tiled into: copy before the next one lands
            out.extend(c.clone() for c in self.ready)
            self.ready.clear()

        if self.cams:
            drain()
            self._launch(stats)
        for j in range(self.K):
            k = (self.launches + j) % self.K
            if self.pending[k] is not None:
                drain()
                self._finish(k, stats)
        drain()
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
