"""Experiment: K frames in flight on one GPU from one host thread (K contexts, each with its own ray pool and
scene replica; rt355_render_tiles_begin / rt355_frame_end).  Throughput against K, for the whole frame and for
the tile share of one rank of 2, 4 and 8."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import Device, scenes, _abi
from raytracer_amd import dist as rd

w, h = int(os.environ.get("W", 1920)), int(os.environ.get("H", 1080))
world, cam = scenes.north_star(w, h)
cd = cam.to_desc()
RANKS = [int(x) for x in os.environ.get("RANKS", "1,2,4,8").split(",")]
KS = [int(x) for x in os.environ.get("KS", "1,2,3,4").split(",")]
BS = [int(x) for x in os.environ.get("BS", "1").split(",")]
for n_ranks, B in [(r, b) for r in RANKS for b in BS]:
    cds = (_abi.CameraDesc * B)(*[cam.to_desc() for _ in range(B)])
    for K in KS:
        devs = [Device(0, 64, flags=_abi.CTX_ONE_STREAM if K > 1 else 0) for _ in range(K)]
        scs = [d.upload(world) for d in devs]
        tpr = rd.tiles_per_rank(w, h, n_ranks)
        bufs = [torch.empty(B * tpr * 1024 * 3, dtype=torch.float32, device="cuda") for _ in range(K)]
        lib = devs[0].lib
        tb = [0.0, 0.0]
        def run(frames):
            inflight = [False] * K
            for f in range(frames):
                k = f % K
                ta = time.perf_counter()
                if inflight[k]:
                    _abi.check(lib.rt355_frame_end(devs[k].handle, None))
                tm = time.perf_counter()
                _abi.check(lib.rt355_render_tiles_batch_begin(devs[k].handle, scs[k].handle, cds, B, 5, 0, n_ranks, C.c_void_p(bufs[k].data_ptr())))
                tb[0] += tm - ta
                tb[1] += time.perf_counter() - tm
                inflight[k] = True
            for k in range(K):
                if inflight[k]:
                    _abi.check(lib.rt355_frame_end(devs[k].handle, None))
        run(6)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 40
        tb[0] = tb[1] = 0.0
        run(n)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / (n * B) * 1e3
        print(f"ranks {n_ranks} (this rank's share), {B} frames per chain, {K} chains in flight: {ms:.3f} ms per frame (host: {tb[0] / (n * B) * 1e3:.3f} ms waiting in frame_end, {tb[1] / (n * B) * 1e3:.3f} ms in begin)", flush=True)
        for s_ in scs: s_.close()
        for d in devs: d.close()
