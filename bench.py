"""Headline benchmark: Mrays/s (primary + secondary + shadow rays) of `Camera::render` on the north-star
workload -- teapot-high (6 320 smooth triangles) on a reflective checker floor next to a glass sphere,
shadows, reflection / refraction depth 5, 1920x1080 -- on N MI355X of one node.

    python bench.py --gpus 1 --steps 240 --warmup 24
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A step is ONE frame: every rank renders its 32x32 tiles (tile t -> rank t % N, scene replicated), the
rank tile buffers are gathered to rank 0 (torch.distributed.gather = RCCL over xGMI) and un-tiled there;
with one rank the canvas is written row-major in place (same kernels, no gather).  Frames are pipelined:
--batch B frames share one launch chain (default: max(16, N), so that a rank's chain carries at least a whole
frame's worth of tiles whatever N is, bounded by 32 GB of ray pool per chain; a run shorter than two such chains is cut
into two equal ones) and --in-flight K (default 3) chains are in flight on every GPU; a step adds one frame and
hands back the oldest finished one, and the timed region ends with a flush, so it holds exactly `steps`
complete frames.
The scene is uploaded once, before the timed region; all inputs are resident in HBM when timing starts.
Every frame of a run has its own camera (a slow orbit around the scene), so no two frames in flight are the same
image.  Rank 0 prints one JSON line.  `value` counts the rays actually traced in the timed region, frame by frame
on the device: one radiance ray per color_at call plus one shadow ray per hit (src/world.rs:58-66, :105), summed
over all ranks; `config.rays_per_frame` is their mean per frame.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from raytracer_amd import Device, _abi, scenes  # noqa: E402
from raytracer_amd import dist as rd  # noqa: E402
from raytracer_amd._abi import Stats  # noqa: E402

FP64_VALU_PEAK_TFLOPS = 78.6  # 256 CU x 4 SIMD x 16 lanes x 2 (FMA) x 2.4 GHz
KIND_NAME = {1: "k_trace", 2: "k_shadow", 3: "k_combine", 5: "k_spawn", 6: "k_scan"}


def cpu_baseline(world, cam, depth, strict_uv=False):
    """The CPU restatement of the reference path (oracle/, float = the reference's precision, brute
    force, recursive, 1 thread like the reference) on a bounded window of the same frame.
    strict_uv: smooth-triangle normals from the u, v of the line's first intersection, as the reference literally does
    (src/intersection.rs:67-71); hits where it would panic are counted and shaded with the hit's own u, v."""
    import oracle
    w, h = int(cam.horizontal_size), int(cam.vertical_size)
    ww, wh = max(1, w // 4), max(1, h // 4)
    x0, y0 = (w - ww) // 2, int(h * 0.55) - wh // 2
    win = (x0, x0 + ww, y0, y0 + wh)
    flags = oracle.STRICT_UV if strict_uv else 0
    _, cnt, sec = oracle.render(world, cam, depth, 32, flags, win, threads=1)
    rays = int(cnt["radiance"].sum() + cnt["hits"].sum())
    # the same window on all the cores this job may use (rows handed out to threads): the reference itself has no
    # threading, this is the restatement's figure, labelled so
    n_thr = max(1, min(16, os.cpu_count() or 1))
    _, _, sec_all = oracle.render(world, cam, depth, 32, flags, win, threads=n_thr)
    return {"value": round(rays / sec / 1e6, 6), "unit": "Mrays/s", "cores": 1, "kind": "port",
            **({"strict_uv": True, "hits_where_the_reference_panics": int(cnt["panics"])} if strict_uv else {}),
            "sample": f"{ww}x{wh} pixel window at ({x0},{y0}) of the same {w}x{h} frame, {rays} rays, "
                      f"{sec:.1f} s, float, brute force over {world.n_shapes()} shapes, 1 thread",
            "all_cores": {"value": round(rays / sec_all / 1e6, 6), "unit": "Mrays/s", "cores": n_thr, "kind": "port (restatement, threads over rows)",
                          "seconds": round(sec_all, 2)}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=240)
    ap.add_argument("--warmup", type=int, default=24)
    ap.add_argument("--scene", default="north_star")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--depth", type=int, default=5)
    ap.add_argument("--real", type=int, default=64, choices=[32, 64])
    ap.add_argument("--in-flight", type=int, default=3, help="frames kept in flight on each GPU (1: every step waits for its own frame)")
    ap.add_argument("--batch", type=int, default=0, help="frames per launch chain (0: max(16, number of GPUs), bounded by 32 GB of ray pool per chain)")
    ap.add_argument("--canvas", default="auto", choices=["auto", "float", "u8"],
                    help="what a step leaves on rank 0: the float canvas (Canvas.pixels) or its Color::to_u8 bytes, the body of the PPM the "
                         "reference writes per frame.  auto: float on one GPU (rendered in place, nothing to gather), u8 on several (the tiles "
                         "are converted where they are rendered and a quarter of the bytes cross xGMI)")
    ap.add_argument("--ref-parser-quirk", action="store_true",
                    help="triangulate the OBJ meshes the way the reference's parser does (polygons fanned over the global vertex list, "
                         "src/wavefront.rs:101-116): the scene the reference itself would build from the file")
    ap.add_argument("--strict-uv-quirk", action="store_true",
                    help="the CPU baseline interpolates smooth-triangle normals with the u, v of the line's first intersection as the "
                         "reference literally does (src/intersection.rs:67-71) and reports the hits where the reference would panic; "
                         "the device always uses the hit's own u, v (declared deviation)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the 3840x2160 figure that is reported beside the headline")
    ap.add_argument("--rehearse", action="store_true",
                    help="all ranks on GPU 0 with the gloo backend: walks the N>1 code path on a one-GPU box (numbers are meaningless)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world_size = int(os.environ.get("WORLD_SIZE", 1))
    if world_size != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_size}: launch with torch.distributed.run")
    if args.rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world_size > 1:
        if args.rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    scenes.REF_PARSER_QUIRK = args.ref_parser_quirk
    world, cam = scenes.CONFIGS[args.scene](args.width, args.height)
    w, h = args.width, args.height
    dev = Device.get(local_rank, args.real)
    scene = dev.upload(world)  # once, outside the timed region
    # K frames are kept in flight on each GPU (one context per frame in flight): every step begins a frame and
    # completes the one begun K steps earlier; the flush before each fence completes the rest, so exactly
    # `steps` whole frames (render + gather + un-tile) lie between the two fences.
    # frames per chain: at least a whole frame's worth of tiles per rank, as long as one chain's ray pool (8 ray slots
    # per primary ray, 116 B each in f64) stays below 32 GB: three chains in flight then hold a third of the 288 GB of HBM.
    # Measured on the headline frame (ms per step, chains in flight x frames per chain): 3 x 8 0.502, 3 x 12 0.485,
    # 3 x 16 0.469, 2 x 24 0.474, 2 x 32 0.478 -- the deeper generations of a chain hold few rays, and every
    # launch ends on its slowest wave.
    pool_bytes_per_frame = rd.tiles_per_rank(w, h, world_size) * rd.TILE * rd.TILE * 8 * (116 if args.real == 64 else 64)
    batch = args.batch if args.batch > 0 else max(1, min(max(16, world_size), int(32e9 // pool_bytes_per_frame)))
    if args.batch <= 0:
        # A run of few frames is cut into equal chains (at least two, so that they overlap) instead of full chains and a
        # short last one: 20 frames as 10 + 10 take 0.433 ms each, as 16 + 4 0.462, as 7 + 7 + 6 0.442.
        n_chains = max(2, -(-args.steps // batch))
        batch = max(1, min(batch, -(-args.steps // n_chains)))
    as_u8 = args.canvas == "u8" or (args.canvas == "auto" and world_size > 1)
    loop = rd.FrameLoop(scene, dev, w, h, rank, world_size, args.depth, in_flight=args.in_flight, batch=batch, as_u8=as_u8)
    local = torch.empty(rd.tiles_per_rank(w, h, world_size) * rd.TILE * rd.TILE * 3, dtype=torch.float32, device="cuda")
    st = Stats()
    # every frame has a camera of its own: a slow orbit around the scene (0.001 rad per frame, 64 frames, then again),
    # so no two frames in flight are the same image; the first camera is the configuration's own
    path = [cam] + [scenes.orbit(cam, 0.001 * k) for k in range(1, 64)]
    n_step = [0]

    def step():
        loop.step(path[n_step[0] % len(path)], st)
        n_step[0] += 1

    def fence():
        torch.cuda.synchronize()
        if world_size > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed set-up: one full chain through every context, so that each has allocated its ray pool
    for _ in range(loop.K * loop.B):
        step()
    loop.flush(st)
    for _ in range(args.warmup):
        step()
    loop.flush(st)
    fence()
    rays_before, n_step[0] = loop.rays_traced, 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    loop.flush(st)
    fence()
    elapsed = time.perf_counter() - t0
    rays_timed = loop.rays_traced - rays_before  # counted by the device, frame by frame
    # one frame alone, begun and waited for (no other frame in flight): the latency beside the rate
    rd.render_tiles(scene, cam, args.depth, rank, world_size, local, st)  # this context's pool is allocated here
    t1 = time.perf_counter()
    for _ in range(5):
        rd.render_tiles(scene, cam, args.depth, rank, world_size, local, st)
    single_ms = (time.perf_counter() - t1) / 5 * 1e3
    primary_local = _active_pixels(w, h, rank, world_size)

    tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    rr = torch.tensor([rays_timed, primary_local], dtype=torch.int64, device="cuda")
    if world_size > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(rr, op=dist.ReduceOp.SUM)
    elapsed = float(tt.item())
    rays_total, primary_frame = int(rr[0].item()), int(rr[1].item())
    rays_frame = rays_total // args.steps  # mean over the frames of the timed region

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        out = {
            "metric": "Mrays/sec (primary+secondary) on " + ("teapot-high 1920x1080" if (args.scene, w, h) == ("north_star", 1920, 1080)
                                                              else f"{args.scene} {w}x{h}"),
            "value": round(rays_total / elapsed / 1e6, 3),
            "unit": "Mrays/s",
            "n_gpus": world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64" if args.real == 64 else "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.scene} {w}x{h} depth {args.depth}: teapot-high.obj (6320 smooth triangles) + "
                                   "reflective checker floor + glass sphere, shadows" if args.scene == "north_star"
                       else f"{args.scene} {w}x{h} depth {args.depth}",
                       "ref_parser_quirk": bool(args.ref_parser_quirk),
                       "shapes": world.n_shapes(), "rays_per_frame": rays_frame, "primary_rays_per_frame": primary_frame,
                       "primary_mrays_per_s": round(primary_frame / (ms * 1e-3) / 1e6, 3),
                       "parallelism": f"tile{world_size}",
                       "canvas": "u8" if as_u8 else "float",
                       # bytes that arrive at rank 0 per frame in the canvas gather
                       "gather_bytes_per_frame": (world_size - 1) * rd.tiles_per_rank(w, h, world_size) * rd.TILE * rd.TILE * 3 * (1 if as_u8 else 4), "chains_in_flight": loop.K, "frames_per_chain": loop.B, "single_frame_ms": round(single_ms, 4),
                       "bvh": scene.info()},
        }
        # --- roofline of the dominant kernel: separate frames, never inside the timed region above ----
        peak64, peak32 = FP64_VALU_PEAK_TFLOPS, 2 * FP64_VALU_PEAK_TFLOPS
        dev.set_timing(True)
        n_prof = 5
        acc = None
        # the chains of the timed loop (loop.B frames each, begun without waiting), one at a time on one stream
        chain_out = torch.empty(loop.B * local.numel(), dtype=torch.float32, device="cuda")
        for _ in range(n_prof):
            rd.render_tiles_chain(scene, [cam] * loop.B, args.depth, rank, world_size, chain_out, st)
            ms_l = np.array(st.launch_ms[: st.n_launch], dtype=np.float64)
            acc = ms_l if acc is None else acc + ms_l
        dev.set_timing(False)
        ms_l = acc / n_prof
        kinds = list(st.launch_kind[: st.n_launch])
        gens = list(st.launch_gen[: st.n_launch])

        def counted(mode):
            dev.set_counting(mode)
            try:
                rd.render_tiles_chain(scene, [cam] * loop.B, args.depth, rank, world_size, chain_out, st)
            finally:
                dev.set_counting(0)
            assert list(st.launch_kind[: st.n_launch]) == kinds and list(st.launch_gen[: st.n_launch]) == gens
            fl = np.array(st.launch_flops[: st.n_launch], dtype=np.float64)
            box = 24.0 * np.array(st.launch_box_tests[: st.n_launch], dtype=np.float64)  # the box tests run in f32
            return fl - box, box, list(st.launch_max_box_tests[: st.n_launch])

        # The numerator is the PER-RAY count: every generation traced one ray per lane over the same BVH, so a ray is
        # charged for the boxes and primitives its own path reaches and not for what the other 63 rays of a packet
        # made its wave visit.  The count of the kernels as they run (packets) is reported beside it.
        f64_ray, f32_ray, maxbox = counted(2)
        f64_run, f32_run, _ = counted(1)
        rays_gen = [int(x) for x in st.radiance_rays]
        hits_gen = [int(x) for x in st.hits]
        k = int(np.argmax(ms_l))
        sec_k = ms_l[k] * 1e-3
        achieved = f64_ray[k] / sec_k / 1e12
        kernel_name = f"{KIND_NAME.get(kinds[k], '?')} generation {gens[k]}"
        headline = (args.scene, w, h, args.depth, args.real, world_size) == ("north_star", 1920, 1080, 5, 64, 1)
        counters = _measured_counters(kernel_name, loop.B) if headline else None
        traffic = int(counters["hbm_bytes"]) if counters else None
        in_f64 = args.real == 64
        out["roofline"] = {
            # arithmetic of the path in the graded precision against that precision's vector peak; the conservative
            # single-precision box tests are reported against the single-precision peak beside it, never summed in
            "bound": "valu_fp64" if in_f64 else "valu_fp32",
            "kernel": kernel_name,
            "achieved": round(achieved if in_f64 else (f64_ray[k] + f32_ray[k]) / sec_k / 1e12, 4),
            "peak": peak64 if in_f64 else peak32,
            "unit": "TFLOP/s",
            "frac": round((achieved / peak64) if in_f64 else (f64_ray[k] + f32_ray[k]) / sec_k / 1e12 / peak32, 5),
            "traffic": traffic,
            "count": "per ray: SURVEY-style algorithmic FLOPs of the boxes / primitives each ray's own BVH path tests (counting mode 2)",
            "flops_per_launch": float(f64_ray[k]),
            "f32_box_tests": {"flops_per_launch": float(f32_ray[k]), "achieved": round(float(f32_ray[k] / sec_k / 1e12), 4), "peak": peak32,
                              "frac": round(float(f32_ray[k] / sec_k / 1e12 / peak32), 5)},
            "as_executed_by_packets": {"flops_per_launch_f64": float(f64_run[k]), "flops_per_launch_f32_boxes": float(f32_run[k]),
                                        "note": "every live lane charged for every node and triangle its wave visits; not used for frac"},
            "launch_ms": round(float(ms_l[k]), 4),
            # a launch works on a chain of this many frames: the per-launch figures above and below are per chain
            "frames_per_launch": loop.B,
            "frame_flops_f64": float(f64_ray.sum() / loop.B),
            "frame_flops_f32_boxes": float(f32_ray.sum() / loop.B),
            "frame_kernel_ms": round(float(ms_l.sum() / loop.B), 4),
            # whole frame against the peak, by the wall time of the timed region (chains overlap)
            "frame_frac": round(float(f64_ray.sum() / loop.B / (ms * 1e-3) / 1e12 / peak64), 5) if in_f64 else
                          round(float((f64_ray + f32_ray).sum() / loop.B / (ms * 1e-3) / 1e12 / peak32), 5),
            "launches": [{"kernel": KIND_NAME.get(kinds[i], "?"), "gen": gens[i], "ms": round(float(ms_l[i]), 4),
                          "gflop_f64": round(float(f64_ray[i]) / 1e9, 4), "gflop_f32_boxes": round(float(f32_ray[i]) / 1e9, 4),
                          "gflop_f64_packets": round(float(f64_run[i]) / 1e9, 4), "gflop_f32_boxes_packets": round(float(f32_run[i]) / 1e9, 4),
                          "max_box_tests_one_ray": int(maxbox[i])} for i in range(len(kinds))],
            "rays_by_generation": rays_gen, "hits_by_generation": hits_gen,
        }
        if counters:
            gbs = traffic / sec_k / 1e9
            out["roofline"]["hbm"] = {"achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4)}
            # the same launch by the hardware's own count of FP64 vector instructions: 64 x (ADD + MUL + TRANS + 2 FMA)
            # is an upper bound on executed FP64 FLOPs (idle lanes included), from the committed PMC passes
            out["roofline"]["counters"] = {k2: counters[k2] for k2 in ("fp64_flops_upper_bound", "lane_utilisation", "wait_any_frac",
                                                                       "valu_active_frac", "l2_hit_rate", "valu_instructions", "source") if k2 in counters}
            out["roofline"]["counters"]["fp64_frac_upper_bound"] = round(counters["fp64_flops_upper_bound"] / sec_k / 1e12 / peak64, 5)
        if world_size == 1:
            out["config"].update(_boundary_call(dev, scene, cam, args.depth, w, h))
        if world_size == 1 and headline and not args.no_extra:
            out["config"]["also"] = _other_sizes(dev, scene, args)
        if not args.no_cpu_baseline and world_size == 1:
            out["cpu_baseline"] = cpu_baseline(world, cam, args.depth, args.strict_uv_quirk)
        print(json.dumps(out))
    if world_size > 1:
        dist.barrier()
        dist.destroy_process_group()


def _boundary_call(dev, scene, cam, depth, w, h):
    """The drop-in call itself, one frame in and one host canvas out like Camera::render (src/camera.rs:74-88):
    rt355_render into pinned host memory (rt355_host_alloc) and rt355_render_u8, wall time per call."""
    lib = dev.lib
    cd = cam.to_desc()
    n = w * h * 3
    out = {}
    for name, fn, nbytes in (("boundary_call_ms", lib.rt355_render, n * 4), ("boundary_call_u8_ms", lib.rt355_render_u8, n)):
        buf = lib.rt355_host_alloc(nbytes)
        if not buf:
            continue
        try:
            for _ in range(2):
                _abi.check(fn(dev.handle, scene.handle, C.byref(cd), depth, C.c_void_p(buf), None))
            t0 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                _abi.check(fn(dev.handle, scene.handle, C.byref(cd), depth, C.c_void_p(buf), None))
            out[name] = round((time.perf_counter() - t0) / reps * 1e3, 4)
        finally:
            lib.rt355_host_free(C.c_void_p(buf))
    return out


def _other_sizes(dev, scene, args):
    """The same scene at 3840x2160 (outside the timed region): where the launch chain stops dominating."""
    w, h = 3840, 2160
    _, cam = scenes.CONFIGS[args.scene](w, h)
    canvas = torch.empty(h * w * 3, dtype=torch.float32, device="cuda")
    st = Stats()
    for _ in range(2):
        rd.render_canvas(scene, cam, args.depth, canvas, st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        rd.render_canvas(scene, cam, args.depth, canvas, st)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    rays = int(st.total_radiance + st.total_shadow) - (int(st.radiance_rays[0]) - w * h)
    return {f"{args.scene} {w}x{h}": {"ms_per_frame": round(ms, 4), "rays_per_frame": rays, "mrays_per_s": round(rays / ms / 1e3, 1)}}


def source_hash():
    """Hash of the device sources: the committed PMC figures are only quoted for the kernels they were measured on."""
    import hashlib
    hsh = hashlib.sha256()
    for name in ("kernels.h", "frame.h", "frame_inst.hip", "rt355.hip", "rt355_device.h", "scene_build.cpp"):
        with open(os.path.join(ROOT, "raytracer_amd", "csrc", name), "rb") as fh:
            hsh.update(fh.read())
    return hsh.hexdigest()[:16]


def _measured_counters(kernel_name, frames_per_chain):
    """Hardware-counter figures of one launch of the named kernel on the headline workload, from the committed PMC
    passes (a bench run cannot collect hardware counters itself).  None when there is no such measurement or
    when it was taken on other kernel sources than the ones in the tree."""
    try:
        with open(os.path.join(ROOT, "profiles", "r03_counters.json")) as fh:
            doc = json.load(fh)
        if doc.get("source_hash") != source_hash():
            return None
        c = dict(doc["launches"][kernel_name])
        c["source"] = "profiles/r03_counters.json"
        measured_on = int(doc.get("frames_per_chain", 0))
        if measured_on != frames_per_chain:
            # The passes ran on chains of another length.  A launch's bytes and instructions are those of its frames' rays
            # (the frames of a chain share nothing but the scene), so the totals go with the number of frames; the
            # ratios stay as measured.
            if measured_on <= 0:
                return None
            for k in ("hbm_bytes", "fp64_flops_upper_bound", "valu_instructions", "fetch_kib", "write_kib", "waves"):
                if k in c:
                    c[k] = c[k] * frames_per_chain / measured_on
            c["source"] += f" (measured on chains of {measured_on} frames, totals scaled to {frames_per_chain})"
        return c
    except (OSError, KeyError, ValueError):
        return None


def _active_pixels(w, h, rank, world_size):
    """Pixels of the frame this rank owns (its tile slots minus padding)."""
    idx = rd.tile_pixels(w, h, rank, world_size)
    return int((idx >= 0).sum())


if __name__ == "__main__":
    main()
