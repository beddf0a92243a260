"""Experiment: one frame rendered as S concurrent strands (tile t -> strand t % S), each strand with a context
(streams, ray pool) of its own, driven by a host thread of its own.  Measures whether the launch chains of
several strands overlap on one GPU."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import Device, scenes
from raytracer_amd import dist as rd

w, h = 1920, 1080
world, cam = scenes.north_star(w, h)
for S in (1, 2, 3, 4, 6, 8):
    devs = [Device(0, 64) for _ in range(S)]
    scs = [d.upload(world) for d in devs]
    tpr = rd.tiles_per_rank(w, h, S)
    bufs = [torch.empty(tpr * 1024 * 3, dtype=torch.float32, device="cuda") for _ in range(S)]
    frames = 30
    barrier = threading.Barrier(S + 1)

    def work(s):
        for f in range(frames + 5):
            barrier.wait()
            rd.render_tiles(scs[s], cam, 5, s, S, bufs[s])
            barrier.wait()

    th = [threading.Thread(target=work, args=(s,)) for s in range(S)]
    [t.start() for t in th]
    ts = []
    for f in range(frames + 5):
        barrier.wait()
        t0 = time.perf_counter()
        barrier.wait()
        ts.append(time.perf_counter() - t0)
    [t.join() for t in th]
    ts = sorted(ts[5:])
    print(f"S={S}: median {ts[len(ts)//2]*1e3:.3f} ms  min {ts[0]*1e3:.3f} ms", flush=True)
    for sc in scs: sc.close()
    for d in devs: d.close()
