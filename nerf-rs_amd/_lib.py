"""Loader for libnerf_mi355x.so + ctypes prototypes of every symbol include/nerf_mi355x.h declares."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

f32p = C.POINTER(C.c_float)
u32p = C.POINTER(C.c_uint32)


class NerfError(RuntimeError):
    """Raised where the reference panics (src/lib.rs:36,118,127,483-501) or the device fails."""

    def __init__(self, code, msg):
        super().__init__(f"[{code}] {msg}")
        self.code = code
        self.msg = msg


class CCamera(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ny", C.c_int32), ("alpha_width", C.c_float), ("alpha_height", C.c_float),
                ("pos", C.c_float * 3), ("dir", C.c_float * 3), ("up", C.c_float * 3),
                ("near", C.c_float), ("far", C.c_float)]


class COpts(C.Structure):
    _fields_ = [("n_coarse", C.c_int32), ("n_fine", C.c_int32), ("coarse_only", C.c_int32),
                ("crop_x0", C.c_int32), ("crop_y0", C.c_int32), ("crop_w", C.c_int32), ("crop_h", C.c_int32),
                ("ssaa", C.c_int32), ("seed", C.c_uint64), ("mlp_dtype", C.c_int32), ("skip_empty", C.c_int32), ("skip_dead", C.c_int32), ("hybrid_sampling", C.c_int32), ("certify_zero", C.c_int32),
                ("band_index", C.c_int32), ("band_count", C.c_int32), ("band_stripe_rows", C.c_int32)]


class CStats(C.Structure):
    _fields_ = [("n_rays", C.c_uint64), ("n_coarse_points", C.c_uint64), ("n_fine_points", C.c_uint64),
                ("ms_total", C.c_double), ("ms_coarse_mlp", C.c_double), ("ms_fine_mlp", C.c_double),
                ("ms_other", C.c_double), ("n_mlp_launches", C.c_uint32), ("n_passes", C.c_uint32),
                ("n_colour_skipped_points", C.c_uint64), ("n_exec_coarse_trunk", C.c_uint64), ("n_exec_fine_trunk", C.c_uint64),
                ("n_exec_colour", C.c_uint64), ("n_hybrid_rays", C.c_uint64), ("n_nonfinite_points", C.c_uint64),
                ("n_certify_audited", C.c_uint64), ("n_certify_violations", C.c_uint64), ("n_certify_retries", C.c_uint32),
                ("n_certify_fallback_rays", C.c_uint32), ("certify_margin", C.c_float * 2), ("certify_headroom", C.c_float * 2),
                ("certify_max_error", C.c_float * 2)]


# name -> (restype, argtypes); kept in sync with include/nerf_mi355x.h (tests/test_host_logic.py::test_abi_exports_every_declared_symbol checks the header)
PROTOTYPES = {
    "nerf_abi_version": (C.c_int, []),
    "nerf_build_variant": (C.c_char_p, []),
    "nerf_abi_struct_sizes": (None, [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "nerf_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "nerf_destroy": (None, [C.c_void_p]),
    "nerf_last_error": (C.c_char_p, [C.c_void_p]),
    "nerf_device_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_char_p, C.c_size_t]),
    "nerf_load_network_dir": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p]),
    "nerf_load_network_tensors": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64),
                                            C.POINTER(f32p)]),
    "nerf_pack_network_dir": (C.c_int, [C.c_char_p, C.c_char_p]),
    "nerf_load_network_blob": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p]),
    "nerf_camera_from_pose": (C.c_int, [f32p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int,
                                        C.POINTER(CCamera)]),
    "nerf_check_network_dir": (C.c_int, [C.c_char_p]),
    "nerf_check_network_blob": (C.c_int, [C.c_char_p]),
    "nerf_debug_pack_network_dir": (C.c_int, [C.c_char_p, f32p, C.c_size_t, f32p, C.c_size_t, C.POINTER(C.c_size_t),
                                              C.POINTER(C.c_size_t)]),
    "nerf_debug_split_bf16x3": (C.c_int, [f32p, C.c_size_t, C.POINTER(C.c_uint16)]),
    "nerf_debug_split_f16x2": (C.c_int, [f32p, C.c_size_t, C.POINTER(C.c_uint16)]),
    "nerf_debug_certify_policy": (C.c_int, [C.c_float, C.c_uint64, C.c_uint64, C.c_float, C.c_float, f32p]),
    "nerf_forward_batch": (C.c_int, [C.c_void_p, C.c_int, f32p, f32p, C.c_size_t, f32p, f32p]),
    "nerf_forward_batch_ex": (C.c_int, [C.c_void_p, C.c_int, C.c_int, f32p, f32p, C.c_size_t, f32p, f32p]),
    "nerf_forward_batch_device": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                            C.c_void_p, C.c_void_p]),
    "nerf_render_image": (C.c_int, [C.c_void_p, C.POINTER(CCamera), C.POINTER(COpts), f32p, C.POINTER(CStats)]),
    "nerf_render_image_device": (C.c_int, [C.c_void_p, C.POINTER(CCamera), C.POINTER(COpts), C.c_void_p, C.c_void_p,
                                           C.POINTER(CStats)]),
    "nerf_render_image_multi": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(CCamera), C.POINTER(COpts), C.c_int, f32p,
                                          C.POINTER(CStats)]),
    "nerf_create_multi": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]),
    "nerf_multi_release": (None, []),
    "nerf_band_rows": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "nerf_kernel_time_query": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64),
                                         C.POINTER(C.c_uint32), C.c_int]),
    "nerf_debug_shader_clock_mhz": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "nerf_camera_from_json": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(CCamera)]),
    "nerf_camera_from_values": (C.c_int, [C.c_float, C.c_float, f32p, f32p, f32p, f32p, C.c_int, C.c_int,
                                          C.POINTER(CCamera)]),
    "nerf_save_ppm": (C.c_int, [C.c_char_p, C.c_int, C.c_int, f32p]),
    "nerf_quantize_rgb8": (None, [f32p, C.c_size_t, C.POINTER(C.c_uint8)]),
    "nerf_quantize_rgba8": (None, [f32p, C.c_size_t, C.POINTER(C.c_uint8)]),
    "nerf_stage_ray_dirs": (C.c_int, [C.c_void_p, C.POINTER(CCamera), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, f32p]),
    "nerf_stage_stratified": (C.c_int, [C.c_void_p, C.POINTER(CCamera), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_uint64, f32p]),
    "nerf_stage_resample": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_float, C.c_uint64, u32p, f32p, f32p,
                                      f32p, f32p, f32p, f32p, f32p]),
    "nerf_stage_hybrid_flags": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_float, C.c_uint64, u32p, f32p, f32p, f32p,
                                          C.c_float, C.POINTER(C.c_uint8), f32p]),
    "nerf_stage_integrate": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_float, f32p, f32p, f32p, f32p, f32p]),
}


def lib_path():
    """The in-tree extension.  NERF_MI355X_LIB names another build of the same library (tuning variants, `make variant`): such a
    build reports a non-empty nerf_build_variant() and is only accepted with NERF_ALLOW_VARIANT=1 (the tools/ scripts set it) --
    timing-only switches make results wrong on purpose, so a variant can never silently stand in for the product."""
    return os.environ.get("NERF_MI355X_LIB") or os.path.join(_HERE, "libnerf_mi355x.so")


def build_native(force=False):
    """Compile the HIP kernels + C ABI for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    args = ["make", "-s", "-C", os.path.join(_HERE, "csrc"), "-j4"]
    if force:
        args.append("-B")
    subprocess.check_call(args)
    return lib_path()


def load_library():
    """Load the HIP extension.  There is NO fallback: a missing library is an error."""
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing: run __graft_entry__.build() (or make -C nerf-rs_amd/csrc); "
                              "nerf-rs_amd has no CPU fallback")
        L = C.CDLL(path)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        tag = L.nerf_build_variant()
        if tag and os.environ.get("NERF_ALLOW_VARIANT") != "1":
            raise ImportError(f"{path} is a tuning variant ({tag.decode()}), not the product build; set NERF_ALLOW_VARIANT=1 for experiments")
        sizes = [C.c_size_t() for _ in range(3)]
        L.nerf_abi_struct_sizes(*[C.byref(x) for x in sizes])
        mine = (C.sizeof(CCamera), C.sizeof(COpts), C.sizeof(CStats))
        if tuple(x.value for x in sizes) != mine:
            raise ImportError(f"{path}: struct layouts {tuple(x.value for x in sizes)} differ from the ctypes mirrors {mine} "
                              "(stale build? run __graft_entry__.build())")
        _LIB = L
    return _LIB


def check(code, ctx=None):
    if code != 0:
        msg = load_library().nerf_last_error(ctx)
        raise NerfError(code, msg.decode() if msg else "unknown error")
