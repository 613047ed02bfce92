"""Host-side mirror of the reference's interface for the hot path, over the C ABI.

Reference names kept (elisabeth96/nerf-rs):
  load_network_from_dir   src/lib.rs:108-174
  Network.forward_batch   src/network.rs:197-237   (points 3 x B SoA, view_dirs B x 3 -> colours B x 3, sigma B)
  Camera / camera_from_samples   src/lib.rs:197-211, 614-645
  render_image            src/lib.rs:474-565      (-> ny x nx x 3 linear RGB, pixel (i, j) at [i, j])
  save_ppm                src/lib.rs:567-580
Errors the reference raises as panics surface as NerfError with the same message text.
"""
import ctypes as C
import json

import numpy as np

from . import _lib
from ._lib import CCamera, COpts, CStats, NerfError, check, f32p, u32p

NET_COARSE, NET_FINE = 0, 1
MLP_F32, MLP_BF16 = 0, 1
_DTYPES = {"f32": 0, "float32": 0, 0: 0, "bf16": 1, "bfloat16": 1, 1: 1, "bf16x3": 2, 2: 2, "f16x2": 3, 3: 3}


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(f32p)


class Stats:
    def __init__(self, c):
        for name, _ in CStats._fields_:
            v = getattr(c, name)
            setattr(self, name, v if isinstance(v, (int, float)) else tuple(v))

    def __repr__(self):
        return "Stats(" + ", ".join(f"{k}={v}" for k, v in self.__dict__.items()) + ")"


class Renderer:
    """One context = one GPU (nerf_create).  Holds the two networks the way render_cli_image does (src/lib.rs:651-652)."""

    def __init__(self, device=0):
        self._L = _lib.load_library()
        h = C.c_void_p()
        check(self._L.nerf_create(int(device), C.byref(h)))
        self.handle = h
        self.device = int(device)
        self.coarse = None
        self.fine = None

    def close(self):
        if getattr(self, "handle", None):
            self._L.nerf_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_info(self):
        n = C.c_int()
        buf = C.create_string_buffer(64)
        check(self._L.nerf_device_info(self.handle, C.byref(n), buf, 64), self.handle)
        return {"n_cus": n.value, "arch": buf.value.decode()}

    def load_scene(self, root):
        """lego_rust/{coarse,fine} (src/lib.rs:650-652)."""
        import os
        self.coarse = load_network_from_dir(self, NET_COARSE, os.path.join(root, "coarse"))
        self.fine = load_network_from_dir(self, NET_FINE, os.path.join(root, "fine"))
        return self.coarse, self.fine

    def kernel_time_query(self, reset=True):
        ms, pts, n = C.c_double(), C.c_uint64(), C.c_uint32()
        check(self._L.nerf_kernel_time_query(self.handle, C.byref(ms), C.byref(pts), C.byref(n), int(reset)), self.handle)
        return ms.value, pts.value, n.value

    # ---- stage entry points (parity tests; hosts that own ray setup) ----
    def stage_ray_dirs(self, cam, x0, y0, w, h, normalize=True):
        out = np.empty((h, w, 3), np.float32)
        check(self._L.nerf_stage_ray_dirs(self.handle, C.byref(cam.c), x0, y0, w, h, int(normalize), _p(out)), self.handle)
        return out

    def stage_stratified(self, cam, x0, y0, w, h, count, seed=0):
        out = np.empty((h, w, count), np.float32)
        check(self._L.nerf_stage_stratified(self.handle, C.byref(cam.c), x0, y0, w, h, count, seed, _p(out)), self.handle)
        return out

    def stage_resample(self, t_coarse, sigma_coarse, nf, far, seed=0, pixel_index=None, u=None):
        t = _f32(t_coarse); s = _f32(sigma_coarse)
        R, nc = t.shape
        w = np.empty((R, nc), np.float32); cdf = np.empty((R, nc - 1), np.float32)
        tn = np.empty((R, nf), np.float32); tf = np.empty((R, nc + nf), np.float32)
        pix = None if pixel_index is None else np.ascontiguousarray(pixel_index, dtype=np.uint32)
        uu = None if u is None else _f32(u)
        check(self._L.nerf_stage_resample(self.handle, R, nc, nf, far, seed,
                                          None if pix is None else pix.ctypes.data_as(u32p), _p(t), _p(s),
                                          None if uu is None else _p(uu), _p(w), _p(cdf), _p(tn), _p(tf)), self.handle)
        return {"w": w, "cdf": cdf, "t_new": tn, "t_fine": tf}

    def stage_hybrid_flags(self, t_coarse, sigma_coarse, nf, far, seed=0, pixel_index=None, u=None, tau=0.0):
        """hybrid_sampling's per-ray decision for the given coarse densities -> (flags (R,) bool, t_new (R, nf) unsorted draws)."""
        t = _f32(t_coarse); s = _f32(sigma_coarse)
        R, nc = t.shape
        flags = np.zeros(R, np.uint8); tn = np.empty((R, nf), np.float32)
        pix = None if pixel_index is None else np.ascontiguousarray(pixel_index, dtype=np.uint32)
        uu = None if u is None else _f32(u)
        check(self._L.nerf_stage_hybrid_flags(self.handle, R, nc, nf, far, seed, None if pix is None else pix.ctypes.data_as(u32p),
                                              _p(t), _p(s), None if uu is None else _p(uu), tau,
                                              flags.ctypes.data_as(C.POINTER(C.c_uint8)), _p(tn)), self.handle)
        return flags.astype(bool), tn

    def stage_integrate(self, rgb, sigma, t, far):
        c = _f32(rgb); s = _f32(sigma); t = _f32(t)
        R, n = t.shape
        out = np.empty((R, 3), np.float32); w = np.empty((R, n), np.float32)
        check(self._L.nerf_stage_integrate(self.handle, R, n, far, _p(c), _p(s), _p(t), _p(out), _p(w)), self.handle)
        return out, w


class Network:
    """network::Network (src/network.rs:172-238) resident on the GPU."""

    def __init__(self, renderer, which):
        self.renderer = renderer
        self.which = which

    def forward_batch(self, points, view_dirs, dtype="f32"):
        """points: (3, B) f32 SoA; view_dirs: (B, 3) -> (colours (B, 3), sigma (B,)).  B == 0 returns empties
        (src/network.rs:199-201)."""
        pts = _f32(points); dirs = _f32(view_dirs)
        if pts.ndim != 2 or pts.shape[0] != 3:
            raise NerfError(-1, "points must be a 3 x B matrix")  # debug_assert_eq!(points.rows(), 3)
        n = pts.shape[1]
        if dirs.shape != (n, 3):
            raise NerfError(-1, "view_dirs must have one direction per column")  # debug_assert_eq!(batch, view_dirs.len())
        rgb = np.empty((n, 3), np.float32); sig = np.empty((n,), np.float32)
        if n:
            R = self.renderer
            check(R._L.nerf_forward_batch_ex(R.handle, self.which, _DTYPES[dtype], _p(pts), _p(dirs), n, _p(rgb), _p(sig)), R.handle)
        return rgb, sig

    def forward_batch_device(self, d_points, d_view_dirs, d_rgb, d_sigma, n, stream=0):
        """Raw device pointers (ints), asynchronous on `stream`."""
        R = self.renderer
        check(R._L.nerf_forward_batch_device(R.handle, self.which, d_points, d_view_dirs, n, d_rgb, d_sigma, stream), R.handle)


def load_network_from_dir(renderer, which, directory):
    """load_network_from_dir (src/lib.rs:108-174): shapes.txt + <name>.bin, assembled by tensor name."""
    check(renderer._L.nerf_load_network_dir(renderer.handle, which, str(directory).encode()), renderer.handle)
    return Network(renderer, which)


def pack_network_dir(directory, blob_path):
    """Convert a reference-format weight directory into the packed one-memcpy blob (host-only)."""
    check(_lib.load_library().nerf_pack_network_dir(str(directory).encode(), str(blob_path).encode()))


def load_network_blob(renderer, which, blob_path):
    check(renderer._L.nerf_load_network_blob(renderer.handle, which, str(blob_path).encode()), renderer.handle)
    return Network(renderer, which)


class Camera:
    """struct Camera (src/lib.rs:197-211)."""

    def __init__(self, c, samples_per_ray=64):
        self.c = c
        self.samples_per_ray = samples_per_ray

    nx = property(lambda self: self.c.nx)
    ny = property(lambda self: self.c.ny)
    near = property(lambda self: self.c.near)
    far = property(lambda self: self.c.far)
    pos = property(lambda self: np.array(list(self.c.pos), np.float32))
    dir = property(lambda self: np.array(list(self.c.dir), np.float32))
    up = property(lambda self: np.array(list(self.c.up), np.float32))


def camera_from_samples(samples, width, height, coarse_samples_per_ray=64):
    """camera_from_samples (src/lib.rs:614-645).  `samples` is the parsed JSON (dict) or a path to it."""
    L = _lib.load_library()
    c = CCamera()
    if isinstance(samples, (str, bytes)) or hasattr(samples, "__fspath__"):
        import os
        check(L.nerf_camera_from_json(os.fspath(samples).encode() if not isinstance(samples, bytes) else samples,
                                      width, height, C.byref(c)))
    else:
        try:
            o = _f32(samples["camera_origin"]); fw = _f32(samples["camera_forward"]); up = _f32(samples["camera_up"])
            hwf = _f32(samples["hwf"]); near = float(samples["near"]); far = float(samples["far"])
        except (KeyError, TypeError, ValueError) as e:
            raise NerfError(-7, f"camera JSON: missing or malformed key {e}")
        check(L.nerf_camera_from_values(near, far, _p(o), _p(fw), _p(up), _p(hwf), width, height, C.byref(c)))
    return Camera(c, coarse_samples_per_ray)


def camera_from_pose(c2w, hwf, near, far, width, height, coarse_samples_per_ray=64):
    """Camera from a 3x4 camera-to-world matrix (the JSON's "camera_matrix") and hwf = (H, W, focal)."""
    m = _f32(np.asarray(c2w)[:3, :4]).reshape(-1)
    c = CCamera()
    check(_lib.load_library().nerf_camera_from_pose(_p(m), float(hwf[0]), float(hwf[1]), float(hwf[2]), float(near), float(far),
                                                  width, height, C.byref(c)))
    return Camera(c, coarse_samples_per_ray)


class RenderOpts:
    def __init__(self, n_coarse=64, n_fine=128, coarse_only=False, crop=None, ssaa=1, seed=0, dtype="f32", skip_empty=False,
                 skip_dead=False, hybrid_sampling=False, certify_zero=False, band=None):
        self.band = tuple(int(v) for v in band) if band else None   # (index, count, stripe_rows): nerf_render_opts.band_*
        self.hybrid_sampling = bool(hybrid_sampling)
        self.certify_zero = bool(certify_zero)
        self.n_coarse, self.n_fine, self.coarse_only, self.crop, self.ssaa, self.seed = \
            n_coarse, n_fine, coarse_only, crop, ssaa, seed
        self.dtype = _DTYPES[dtype]
        self.skip_empty = bool(skip_empty)
        self.skip_dead = bool(skip_dead)

    def to_c(self):
        o = COpts()
        o.n_coarse, o.n_fine, o.coarse_only = self.n_coarse, self.n_fine, int(self.coarse_only)
        if self.crop:
            o.crop_x0, o.crop_y0, o.crop_w, o.crop_h = self.crop
        o.ssaa, o.seed, o.mlp_dtype, o.skip_empty = self.ssaa, self.seed, self.dtype, int(self.skip_empty)
        o.skip_dead = int(self.skip_dead)
        o.hybrid_sampling = int(self.hybrid_sampling)
        o.certify_zero = int(self.certify_zero)
        if self.band:
            o.band_index, o.band_count, o.band_stripe_rows = self.band
        return o

    def out_shape(self, cam):
        h, w = (self.crop[3], self.crop[2]) if self.crop else (cam.ny, cam.nx)
        if self.band and self.band[1] > 1:
            h = band_rows(h, *self.band)
        return (h, w, 3)


def band_rows(window_rows, index, count, stripe_rows=0):
    """Rows of band `index` of `count` (nerf_band_rows): stripe_rows = 0 contiguous bands, > 0 stripes of that many rows round-robin."""
    n = _lib.load_library().nerf_band_rows(int(window_rows), int(index), int(count), int(stripe_rows))
    if n < 0:
        raise NerfError(n, "band_index / band_count / band_stripe_rows out of range")
    return n


def band_row_indices(window_rows, index, count, stripe_rows=0):
    """The window rows band `index` holds, in the order it holds them (the layout nerf_render_opts.band_* documents)."""
    h, n = int(window_rows), max(int(count), 1)
    if n == 1:
        return np.arange(h)
    if stripe_rows <= 0:
        base, rem = divmod(h, n)
        y0 = index * base + min(index, rem)
        return np.arange(y0, y0 + base + (1 if index < rem else 0))
    rows = np.arange(h)
    return rows[(rows // stripe_rows) % n == index]


def render_image(coarse, fine, camera, fine_samples_per_ray=128, *, seed=0, coarse_only=False, crop=None, ssaa=1,
                 dtype="f32", skip_empty=False, skip_dead=False, hybrid_sampling=False, certify_zero=False, band=None, return_stats=False,
                 device_out=None, stream=0):
    """render_image (src/lib.rs:474-565) -> (h, w, 3) float32 linear RGB.

    band = (index, count, stripe_rows): only that band of the window's rows, packed (nerf_render_opts.band_*).

    coarse/fine: Network objects of one Renderer; camera.samples_per_ray is the coarse sample count.
    device_out: optional raw device pointer (int) to receive the image instead of a host array (asynchronous on
    `stream`; returns None / stats)."""
    R = coarse.renderer
    if fine is not None and fine.renderer is not R:
        raise NerfError(-1, "coarse and fine networks must live in the same Renderer")
    opts = RenderOpts(camera.samples_per_ray, fine_samples_per_ray, coarse_only, crop, ssaa, seed, dtype, skip_empty, skip_dead, hybrid_sampling, certify_zero, band)
    o = opts.to_c()
    st = CStats()
    if device_out is not None:
        check(R._L.nerf_render_image_device(R.handle, C.byref(camera.c), C.byref(o), device_out, stream,
                                            C.byref(st) if return_stats else None), R.handle)
        return Stats(st) if return_stats else None
    shape = opts.out_shape(camera)
    if shape[0] <= 0 or shape[1] <= 0:
        raise NerfError(-1, "this band has no rows (more bands than rows)" if opts.band and opts.band[1] > 1 and shape[1] > 0 and
                        (crop[3] if crop else camera.ny) > 0 else "crop window outside the frame")
    out = np.empty(shape, np.float32)
    check(R._L.nerf_render_image(R.handle, C.byref(camera.c), C.byref(o), _p(out), C.byref(st)), R.handle)
    return (out, Stats(st)) if return_stats else out


GATHER_HOST, GATHER_PEER, GATHER_RCCL = 0, 1, 2
_GATHERS = {"host": 0, "peer": 1, "rccl": 2, 0: 0, 1: 1, 2: 2}


def render_image_multi(renderers, camera, fine_samples_per_ray=128, *, gather="host", seed=0, coarse_only=False, crop=None,
                       ssaa=1, dtype="f32", skip_empty=False, skip_dead=False, hybrid_sampling=False, certify_zero=False, return_stats=False):
    """render_image fanned out over several Renderers (one per GPU) inside ONE process, through nerf_render_image_multi:
    row bands on per-context host threads + streams, gathered by direct D2H ("host"), GPU-to-GPU peer copies ("peer") or one
    RCCL all-gather ("rccl").  The reference's counterpart is the rayon fan-out + scatter of src/lib.rs:533-557.
    Every Renderer must have both networks loaded."""
    L = _lib.load_library()
    n = len(renderers)
    handles = (C.c_void_p * n)(*[r.handle for r in renderers])
    opts = RenderOpts(camera.samples_per_ray, fine_samples_per_ray, coarse_only, crop, ssaa, seed, dtype, skip_empty, skip_dead, hybrid_sampling, certify_zero)
    o = opts.to_c()
    shape = opts.out_shape(camera)
    if shape[0] <= 0 or shape[1] <= 0:
        raise NerfError(-1, "crop window outside the frame")
    out = np.empty(shape, np.float32)
    st = (CStats * n)()
    check(L.nerf_render_image_multi(handles, n, C.byref(camera.c), C.byref(o), _GATHERS[gather], _p(out), st if return_stats else None),
          renderers[0].handle if n else None)
    return (out, [Stats(s) for s in st]) if return_stats else out


def quantize_rgb8(pixels):
    a = _f32(pixels)
    out = np.empty(a.shape, np.uint8)
    _lib.load_library().nerf_quantize_rgb8(_p(a), a.size // 3, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def quantize_rgba8(pixels):
    """pixels_to_rgba (src/lib.rs:582-592): (..., 3) f32 -> (..., 4) u8 with alpha 255."""
    a = _f32(pixels)
    out = np.empty(a.shape[:-1] + (4,), np.uint8)
    _lib.load_library().nerf_quantize_rgba8(_p(a), a.size // 3, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def save_ppm(path, width, height, pixels):
    """save_ppm (src/lib.rs:567-580); pixels: (height, width, 3) or (height*width, 3)."""
    a = _f32(pixels)
    if a.size != width * height * 3:
        raise NerfError(-1, "pixels.len() != width * height")  # assert_eq! src/lib.rs:569
    check(_lib.load_library().nerf_save_ppm(str(path).encode(), width, height, _p(a)))


def load_tf_samples(path):
    """load_tf_samples (src/lib.rs:94-99)."""
    with open(path) as f:
        return json.load(f)
