"""ctypes view of include/rt355.h and the loader of the HIP backend library.

The library is built in-tree by ``__graft_entry__.build()`` (``raytracer_amd/csrc/librt355.so``).
There is no CPU fallback: if the library is missing, ``lib()`` raises.
"""
import ctypes as C
import os

RT355_MAX_GEN = 8
RT355_MAX_LAUNCH = 48

# enum rt355_kind (order of `enum Shape`, src/shape.rs:11-20)
SPHERE, PLANE, CUBE, CYLINDER, CONE, GROUP, TRIANGLE, SMOOTH_TRIANGLE = range(8)
# enum rt355_pattern (order of `enum Pattern`, src/pattern.rs:6-12; 0 = None)
PATTERN_NONE, PATTERN_STRIPE, PATTERN_GRADIENT, PATTERN_RING, PATTERN_CHECKER, PATTERN_TEST = range(6)

K_TRACE, K_SHADE, K_COMBINE, K_SPAWN, K_SCAN = 1, 2, 3, 5, 6
CTX_ONE_STREAM = 1
MAX_BATCH = 64


class MaterialDesc(C.Structure):
    _fields_ = [
        ("color", C.c_float * 3),
        ("ambient", C.c_float),
        ("diffuse", C.c_float),
        ("specular", C.c_float),
        ("shininess", C.c_float),
        ("reflective", C.c_float),
        ("refractive", C.c_float),
        ("transparency", C.c_float),
        ("pattern_kind", C.c_int32),
        ("c1", C.c_float * 3),
        ("c2", C.c_float * 3),
    ]


class ShapeDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("m", C.c_float * 16),
        ("mat", MaterialDesc),
        ("min", C.c_float),
        ("max", C.c_float),
        ("closed", C.c_int32),
        ("p1", C.c_float * 3),
        ("p2", C.c_float * 3),
        ("p3", C.c_float * 3),
        ("n1", C.c_float * 3),
        ("n2", C.c_float * 3),
        ("n3", C.c_float * 3),
    ]


class LightDesc(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("intensity", C.c_float * 3)]


class CameraDesc(C.Structure):
    _fields_ = [("hsize", C.c_float), ("vsize", C.c_float), ("fov", C.c_float), ("transform", C.c_float * 16)]


class Stats(C.Structure):
    _fields_ = [
        ("radiance_rays", C.c_uint64 * RT355_MAX_GEN),
        ("hits", C.c_uint64 * RT355_MAX_GEN),
        ("total_radiance", C.c_uint64),
        ("total_shadow", C.c_uint64),
        ("frame_ms", C.c_double),
        ("n_launch", C.c_int32),
        ("launch_kind", C.c_int32 * RT355_MAX_LAUNCH),
        ("launch_gen", C.c_int32 * RT355_MAX_LAUNCH),
        ("launch_ms", C.c_float * RT355_MAX_LAUNCH),
        ("launch_box_tests", C.c_uint64 * RT355_MAX_LAUNCH),
        ("launch_prim_tests", C.c_uint64 * RT355_MAX_LAUNCH),
        ("launch_max_box_tests", C.c_uint64 * RT355_MAX_LAUNCH),
        ("launch_max_wave_clocks", C.c_uint64 * RT355_MAX_LAUNCH),
        ("launch_flops", C.c_double * RT355_MAX_LAUNCH),
    ]


# RT355_LIB: another build of the same library (kernel experiments); the default is the in-tree build
LIB_PATH = os.environ.get("RT355_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "librt355.so")

# (name, restype, argtypes): every symbol include/rt355.h declares
SYMBOLS = [
    ("rt355_last_error", C.c_char_p, []),
    ("rt355_version", C.c_int, []),
    ("rt355_ctx_create", C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int]),
    ("rt355_ctx_create_ex", C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int]),
    ("rt355_ctx_destroy", None, [C.c_void_p]),
    ("rt355_ctx_set_pool", C.c_int, [C.c_void_p, C.c_double]),
    ("rt355_ctx_set_exact_cones", C.c_int, [C.c_void_p, C.c_int]),
    ("rt355_ctx_set_counting", C.c_int, [C.c_void_p, C.c_int]),
    ("rt355_ctx_set_timing", C.c_int, [C.c_void_p, C.c_int]),
    ("rt355_scene_upload", C.c_int, [C.c_void_p, C.POINTER(ShapeDesc), C.c_size_t, C.POINTER(LightDesc), C.POINTER(C.c_void_p)]),
    ("rt355_scene_destroy", None, [C.c_void_p]),
    ("rt355_scene_info", C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    ("rt355_render", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(CameraDesc), C.c_int, C.c_void_p, C.POINTER(Stats)]),
    ("rt355_render_u8", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(CameraDesc), C.c_int, C.c_void_p, C.POINTER(Stats)]),
    ("rt355_host_alloc", C.c_void_p, [C.c_size_t]),
    ("rt355_host_free", None, [C.c_void_p]),
    ("rt355_multi_create", C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_int]),
    ("rt355_multi_destroy", None, [C.c_void_p]),
    ("rt355_multi_upload", C.c_int, [C.c_void_p, C.POINTER(ShapeDesc), C.c_size_t, C.POINTER(LightDesc)]),
    ("rt355_multi_render", C.c_int, [C.c_void_p, C.POINTER(CameraDesc), C.c_int, C.c_void_p, C.POINTER(Stats)]),
    ("rt355_multi_render_batch", C.c_int, [C.c_void_p, C.POINTER(CameraDesc), C.c_int, C.c_int, C.c_void_p, C.POINTER(Stats)]),
    ("rt355_multi_gather_kind", C.c_char_p, [C.c_void_p]),
    ("rt355_render_device", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(CameraDesc), C.c_int, C.c_void_p, C.POINTER(Stats)]),
    ("rt355_render_device_noalloc", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(CameraDesc), C.c_int, C.c_void_p, C.POINTER(Stats)]),
    ("rt355_render_tiles", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(CameraDesc), C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(Stats)]),
    ("rt355_render_tiles_begin", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(CameraDesc), C.c_int, C.c_int, C.c_int, C.c_void_p]),
    ("rt355_render_device_begin", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(CameraDesc), C.c_int, C.c_void_p]),
    ("rt355_frame_end", C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    ("rt355_render_tiles_batch_begin", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(CameraDesc), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    ("rt355_render_device_batch_begin", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(CameraDesc), C.c_int, C.c_int, C.c_void_p]),
    ("rt355_render_batch", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(CameraDesc), C.c_int, C.c_int, C.c_void_p, C.POINTER(Stats)]),
    ("rt355_tiles_per_rank", C.c_int64, [C.POINTER(CameraDesc), C.c_int]),
    ("rt355_untile", C.c_int, [C.c_void_p, C.POINTER(CameraDesc), C.c_int, C.c_void_p, C.c_void_p]),
    ("rt355_untile_u8", C.c_int, [C.c_void_p, C.POINTER(CameraDesc), C.c_int, C.c_void_p, C.c_void_p]),
    ("rt355_untile_batch", C.c_int, [C.c_void_p, C.POINTER(CameraDesc), C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    ("rt355_untile_batch_u8", C.c_int, [C.c_void_p, C.POINTER(CameraDesc), C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    ("rt355_color_at", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.POINTER(Stats)]),
    ("rt355_trace_sample", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    ("rt355_is_shadowed", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    ("rt355_to_u8", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    ("rt355_to_u8_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    ("rt355_ctx_wait_stream", C.c_int, [C.c_void_p, C.c_void_p]),
    ("rt355_stream_wait_ctx", C.c_int, [C.c_void_p, C.c_void_p]),
    ("rt355_ctx_set_async", C.c_int, [C.c_void_p, C.c_int]),
    ("rt355_ctx_sync", C.c_int, [C.c_void_p]),
    ("rt355_scene_check", C.c_int, [C.POINTER(ShapeDesc), C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_uint64)]),
    ("rt355_flop_table", None, [C.POINTER(C.c_double)]),
]

_lib = None


def lib():
    """Load librt355.so and type every entry point.  Raises if the extension was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"HIP backend not built: {LIB_PATH} is missing. Run `python -c 'import __graft_entry__ as g; g.build()'`."
        )
    # PyTorch wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).  Two HIP runtimes in
    # one process do not work, and the one that is loaded first wins the SONAME: import torch first so
    # that this library binds to torch's runtime whenever both live in one process.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    l = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(l, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = l
    return l


class Rt355Error(RuntimeError):
    pass


def check(code):
    if code != 0:
        msg = lib().rt355_last_error()
        raise Rt355Error(f"rt355 error {code}: {msg.decode() if msg else ''}")
