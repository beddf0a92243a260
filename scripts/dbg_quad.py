"""Debug: determinism of the quad traversal (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracer_amd import Device, scenes
from raytracer_amd import dist as rd
from raytracer_amd._abi import Stats

dev = Device.get(0, 64)
for name, build in [("glass", lambda: scenes.glass(320, 200)), ("stress_mid", lambda: scenes.stress(640, 360, n_top=2, n_mid=20, n_leaf=100, extent=120.0))]:
    world, cam = build()
    scene = dev.upload(world)
    w, h = int(cam.horizontal_size), int(cam.vertical_size)
    a = scene.render(cam, 5).pixels.copy()
    b = scene.render(cam, 5).pixels.copy()
    print(name, "run-to-run differing pixels:", int((np.abs(a - b).max(axis=2) > 0).sum()))
    for n_ranks in (2, 3):
        tpr = rd.tiles_per_rank(w, h, n_ranks)
        g = torch.zeros((n_ranks, tpr * 1024 * 3), dtype=torch.float32, device="cuda")
        for r in range(n_ranks):
            rd.render_tiles(scene, cam, 5, r, n_ranks, g[r])
        out = rd.untile_host(g.cpu().numpy(), w, h, n_ranks)
        d = np.abs(out - a).max(axis=2)
        ys, xs = np.nonzero(d > 0)
        print(name, n_ranks, "ranks: differing pixels", len(ys), "max diff", d.max(), list(zip(xs[:5], ys[:5])))
    # nearest hit: same rays, two orders
    rng = np.random.default_rng(1)
    rays = np.concatenate([rng.uniform(-20, 20, (40000, 3)), rng.normal(size=(40000, 3))], axis=1).astype(np.float32)
    rays[:, 1] = np.abs(rays[:, 1]) * 0.3 + 0.5
    t1, i1 = scene.trace(rays)
    perm = rng.permutation(len(rays))
    t2, i2 = scene.trace(rays[perm])
    same = (i1[perm] == i2) & ((t1[perm] == t2) | (np.isnan(t1[perm]) & np.isnan(t2)))
    print(name, "trace order dependence: differing rays", int((~same).sum()), "hits", int((i1 >= 0).sum()))
    os.environ["X"] = "1"
    scene.close()
