import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from raytracer_amd import Device, scenes
sys.path.insert(0, 'tests')
from test_gpu_parity import rand_rays
dev = Device.get(0, 64)
world, _ = scenes.stress(64, 36, n_top=2, n_mid=6, n_leaf=25, extent=40.0)
rays = rand_rays(30000, 11, center=(0, 1.5, 0), radius=45.0, spread=18.0)
scene = dev.upload(world)
print(scene.info())
t, idx = scene.trace(rays)
to, io, gap = oracle.hit(world, rays, 64)
bad = np.where((idx != io) & (gap >= 1e-6))[0]
print("n bad", len(bad))
kinds = world.records["kind"]
for b in bad[:20]:
    print(b, "dev", idx[b], kinds[idx[b]] if idx[b] >= 0 else None, t[b], "oracle", io[b], kinds[io[b]] if io[b] >= 0 else None, to[b], "gap", gap[b])
    print("   ray", rays[b], "rec", world.records["min"][io[b]], world.records["max"][io[b]], world.records["closed"][io[b]])
