"""CPU tests of the product's host side through the C ABI: exported symbols vs the header, loader failure cases,
camera-from-JSON, PPM writer, and the packed weight layout (emulated lane-by-lane in numpy against the oracle).
No device call is made here (no GPU in the build container)."""
import ctypes as C
import os
import re
import shutil

import numpy as np
import pytest

from conftest import ROOT, SCENE

HEADER = os.path.join(ROOT, "include", "nerf_mi355x.h")


def test_abi_exports_every_declared_symbol(native):
    """Every function include/nerf_mi355x.h declares is exported by the .so and has a ctypes prototype."""
    from nerf_rs_amd import _lib
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(nerf_[a-z0-9_]+)\s*\(", text))
    declared -= {"nerf_ctx"}
    assert len(declared) >= 20
    L = C.CDLL(native.lib_path())
    for name in declared:
        assert hasattr(L, name), f"{name} declared in the header but not exported"
        assert name in _lib.PROTOTYPES, f"{name} has no ctypes prototype"
    assert set(_lib.PROTOTYPES) == declared
    assert native.load_library().nerf_abi_version() == 5
    assert native.load_library().nerf_build_variant() == b""      # the product build; variants carry a tag and are refused by the loader


def test_no_device_means_loud_failure(native):
    """Without a HIP device nerf_create must fail (no CPU fallback).  Skipped where a GPU exists."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(native.NerfError) as e:
        native.Renderer(0)
    assert "no CPU fallback" in str(e.value) or "HIP" in str(e.value)


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under nerf-rs_amd/ may reference it."""
    pkg = os.path.join(ROOT, "nerf-rs_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                src = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle_py" not in src and "nerf_oracle" not in src and "libnerf_oracle" not in src, (dp, f)


def _check(native, d):
    return native.load_library().nerf_check_network_dir(str(d).encode()), \
        (native.load_library().nerf_last_error(None) or b"").decode()


def test_loader_failure_cases(native, tmp_path):
    """load_network_from_dir's failure cases (src/lib.rs:36, 65, 118, 127) as status codes + messages."""
    assert _check(native, os.path.join(SCENE, "coarse"))[0] == 0
    assert _check(native, os.path.join(SCENE, "fine"))[0] == 0
    rc, msg = _check(native, tmp_path / "nope")
    assert rc == -2 and "read shapes" in msg
    d = tmp_path / "net"; shutil.copytree(os.path.join(SCENE, "coarse"), d)
    os.remove(d / "dense3_kernel.bin")
    rc, msg = _check(native, d)
    assert rc == -2 and "read tensor" in msg and "dense3_kernel" in msg
    shutil.copy(os.path.join(SCENE, "coarse", "dense3_kernel.bin"), d / "dense3_kernel.bin")
    lines = [l for l in open(d / "shapes.txt") if not l.startswith("alpha_bias")]
    open(d / "shapes.txt", "w").writelines(lines)
    rc, msg = _check(native, d)
    assert rc == -3 and msg == "missing bias parameter: alpha_bias"
    lines = [l for l in lines if not l.startswith("rgb_kernel")]
    open(d / "shapes.txt", "w").writelines(lines)
    rc, msg = _check(native, d)
    assert rc == -3 and msg == "missing matrix parameter: rgb_kernel"
    shutil.copy(os.path.join(SCENE, "coarse", "shapes.txt"), d / "shapes.txt")
    with open(d / "dense0_kernel.bin", "ab") as f:
        f.write(b"\0" * 8)
    rc, msg = _check(native, d)
    assert rc == -4 and "dims mismatch for dense0_kernel" in msg
    # an unused extra tensor is tolerated (only a debug_assert in the reference, src/lib.rs:171)
    shutil.copy(os.path.join(SCENE, "coarse", "dense0_kernel.bin"), d / "dense0_kernel.bin")
    np.zeros(4, np.float32).tofile(d / "extra.bin")
    with open(d / "shapes.txt", "a") as f:
        f.write("extra 4\n")
    assert _check(native, d)[0] == 0


def test_camera_from_json_matches_oracle_bitwise(native, oracle, samples):
    for w, h in ((256, 256), (400, 400), (800, 600)):
        a = native.camera_from_samples(os.path.join(SCENE, "tf_reference_samples.json"), w, h).c
        b = native.camera_from_samples(samples, w, h).c
        o = oracle.camera_from_samples(samples, w, h)
        for cam in (a, b):
            assert (cam.nx, cam.ny) == (o.nx, o.ny) == (w, h)
            for f in ("alpha_width", "alpha_height", "near", "far"):
                assert np.float32(getattr(cam, f)).tobytes() == np.float32(getattr(o, f)).tobytes()
            for f in ("pos", "dir", "up"):
                assert list(getattr(cam, f)) == list(getattr(o, f))
    assert abs(np.tan(a.alpha_width) - 0.36) < 1e-6 and (a.near, a.far) == (2.0, 6.0)


def test_camera_json_errors(native, tmp_path):
    L = native.load_library()
    from nerf_rs_amd._lib import CCamera
    cam = CCamera()
    assert L.nerf_camera_from_json(str(tmp_path / "missing.json").encode(), 8, 8, C.byref(cam)) == -2
    p = tmp_path / "bad.json"; p.write_text('{"near": 2.0, "far": 6.0, "hwf": [400, 400')
    assert L.nerf_camera_from_json(str(p).encode(), 8, 8, C.byref(cam)) == -7
    p.write_text('{"near": 2.0, "far": 6.0, "hwf": [400, 400, 555.5], "camera_origin": [0,0,0], "camera_up": [0,0,1]}')
    assert L.nerf_camera_from_json(str(p).encode(), 8, 8, C.byref(cam)) == -7
    assert b"camera_forward" in L.nerf_last_error(None)
    with pytest.raises(native.NerfError):
        native.camera_from_samples({"near": 2.0}, 8, 8)


def test_save_ppm_matches_oracle(native, oracle, tmp_path):
    rng = np.random.default_rng(3)
    img = rng.uniform(-0.2, 1.2, size=(5, 7, 3)).astype(np.float32)
    img[0, 0] = [np.nan, 0.5, 1.0]
    assert np.array_equal(native.quantize_rgb8(img), oracle.quantize_rgb8(img))
    rgba = native.quantize_rgba8(img)                                              # pixels_to_rgba, src/lib.rs:582-592
    assert rgba.shape == img.shape[:-1] + (4,) and np.array_equal(rgba[..., :3], native.quantize_rgb8(img)) and (rgba[..., 3] == 255).all()
    a, b = tmp_path / "a.ppm", tmp_path / "b.ppm"
    native.save_ppm(a, 7, 5, img); oracle.save_ppm(b, img)
    assert a.read_bytes() == b.read_bytes() and a.read_bytes().startswith(b"P6\n7 5\n255\n")
    with pytest.raises(native.NerfError):
        native.save_ppm(a, 6, 5, img)  # assert_eq!(pixels.len(), width * height), src/lib.rs:569


# ---- packed weight layout: numpy emulation of the kernel's lane/register layout (mlp_layout.h) --------------------
def _pack(native, which):
    L = native.load_library()
    nw, ns = C.c_size_t(), C.c_size_t()
    d = os.path.join(SCENE, which).encode()
    assert L.nerf_debug_pack_network_dir(d, None, 0, None, 0, C.byref(nw), C.byref(ns)) == 0
    ws = np.empty(nw.value, np.float32); sm = np.empty(ns.value, np.float32)
    f32p = C.POINTER(C.c_float)
    assert L.nerf_debug_pack_network_dir(d, ws.ctypes.data_as(f32p), ws.size, sm.ctypes.data_as(f32p), sm.size, None, None) == 0
    return ws, sm


LANE = np.arange(64); P_ = LANE & 31; H_ = LANE >> 5
ROW_OF = np.array([[(r & 3) + 8 * (r >> 2) + 4 * h for h in (0, 1)] for r in range(16)])  # [r][h]


def _mfma(acc, a, b):
    """v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5] = a[l], B[k=l>>5][j=l&31] = b[l]; D reg r of lane l =
    D[(r&3)+8(r>>2)+4(l>>5)][l&31]."""
    D = a.reshape(2, 32).T.astype(np.float64) @ b.reshape(2, 32).astype(np.float64)
    acc += D[ROW_OF[:, H_], P_[None, :]]


def _emulate_wave(ws, sm, pts, dirs):
    """One 32-point wave tile through the full network exactly as mlp_kernel.hip walks the stream."""
    BIAS, BIASV, AW, RW, MISC = 0, 9 * 256, 9 * 256 + 128, 9 * 256 + 128 + 256, 9 * 256 + 128 + 256 + 384
    pos = pts[:, P_]; d = dirs[P_].T
    E = np.zeros((2, 16, 64))
    f0 = np.where(H_ == 1, 32.0, 1.0)
    for o in range(5):
        for ax in range(3):
            arg = np.float32(f0 * 2.0 ** o) * pos[ax]
            for idx, val in ((6 * o + ax, np.sin(np.float64(arg))), (6 * o + 3 + ax, np.cos(np.float64(arg)))):
                E[idx >> 4, idx & 15] = val
    E[1, 14] = np.where(H_ == 1, pos[2], pos[0]); E[1, 15] = np.where(H_ == 1, 0.0, pos[1])
    cur = [0]

    def bias(off, nt):
        return np.stack([sm[off + (t * 2 + H_) * 16 + r] for t in range(nt) for r in range(16)]).reshape(nt, 16, 64).astype(np.float64)

    def steps(inp, out, relu):  # one input tile (16 k-steps)
        nt = out.shape[0]
        for r in range(16):
            b = np.maximum(inp[r], 0) if relu else inp[r]
            for g in range(nt // 4):
                piece = ws[cur[0]: cur[0] + 256].reshape(64, 4); cur[0] += 256
                for q in range(4):
                    _mfma(out[4 * g + q], piece[:, q], b)

    X = bias(BIAS, 8); steps(E[0], X, False); steps(E[1], X, False)
    for layer in range(1, 5):
        Y = bias(BIAS + layer * 256, 8)
        for t in range(8):
            steps(X[t], Y, True)
        X = Y
    Y = bias(BIAS + 5 * 256, 8); steps(E[0], Y, False); steps(E[1], Y, False)
    for t in range(8):
        steps(X[t], Y, True)
    X = Y
    for layer in (6, 7):
        Y = bias(BIAS + layer * 256, 8)
        for t in range(8):
            steps(X[t], Y, True)
        X = Y
    assert cur[0] == 120 * 4096
    aw = np.stack([sm[AW + H_ * 128 + k] for k in range(128)]).reshape(8, 16, 64)
    part = (aw * np.maximum(X, 0)).sum(axis=(0, 1))
    sigma = np.maximum(part + part[LANE ^ 32] + sm[MISC], 0)
    B = bias(BIAS + 8 * 256, 8)
    for t in range(8):
        steps(X[t], B, True)
    D = np.zeros((16, 64))
    f = np.where(H_ == 1, 4.0, 1.0)
    for o in range(2):
        for ax in range(3):
            arg = np.float32(f * 2.0 ** o) * d[ax]
            D[6 * o + ax] = np.sin(np.float64(arg)); D[6 * o + 3 + ax] = np.cos(np.float64(arg))
    for ax in range(3):
        D[12 + ax] = np.where(H_ == 1, 0.0, d[ax])
    V = bias(BIASV, 4)
    for t in range(8):
        steps(B[t], V, False)
    steps(D, V, False)
    assert cur[0] == 145 * 4096 == ws.size
    rgb = np.zeros((3, 64))
    for c in range(3):
        rw = np.stack([sm[RW + (H_ * 3 + c) * 64 + k] for k in range(64)]).reshape(4, 16, 64)
        part = (rw * np.maximum(V, 0)).sum(axis=(0, 1))
        rgb[c] = 1.0 / (1.0 + np.exp(-(part + part[LANE ^ 32] + sm[MISC + 1 + c])))
    return rgb[:, :32].T, sigma[:32]


@pytest.mark.parametrize("which", ["coarse", "fine"])
def test_packed_layout_reproduces_the_network(native, oracle, samples, oracle_nets, which):
    ws, sm = _pack(native, which)
    assert ws.size == 145 * 4096 and sm.size % 64 == 0
    origin = np.float32(samples["camera_origin"]); z = np.float32(samples["z_vals"])
    pts = np.zeros((3, 32), np.float32); dirs = np.zeros((32, 3), np.float32); dirs[:, 2] = 1
    exp_s, exp_c = [], []
    for e, ex in enumerate(samples["examples"]):
        rd = np.float32(ex["ray_d"])
        pts[:, 5 * e: 5 * e + 5] = origin[:, None] + rd[:, None] * z[None, :]
        dirs[5 * e: 5 * e + 5] = np.float32(ex["viewdir_unit"])
        exp_s += ex[f"{which}_sigma"]; exp_c += ex[f"{which}_rgb"]
    rng = np.random.default_rng(1)
    pts[:, 15:] = rng.uniform(-2.2, 2.2, size=(3, 17)); v = rng.normal(size=(17, 3))
    dirs[15:] = v / np.linalg.norm(v, axis=1, keepdims=True)
    rgb, sigma = _emulate_wave(ws, sm, pts, dirs)
    exp_s, exp_c = np.float32(exp_s), np.float32(exp_c)
    assert np.all(np.abs(sigma[:15] - exp_s) <= 1e-4 * (1 + np.abs(exp_s)))      # the reference's golden scalars
    assert np.all(np.abs(rgb[:15] - exp_c) <= 1e-5)
    o_rgb, o_sig = oracle_nets[0 if which == "coarse" else 1].forward_batch(pts, dirs)
    assert np.all(np.abs(sigma - o_sig) <= 1e-4 * (1 + np.abs(o_sig))) and np.all(np.abs(rgb - o_rgb) <= 1e-5)


def test_camera_from_pose_equals_camera_from_samples(native, samples):
    """The JSON's camera_matrix (unused by the reference) gives the same camera as forward/up/origin."""
    a = native.camera_from_samples(samples, 400, 400).c
    b = native.camera_from_pose(samples["camera_matrix"], samples["hwf"], samples["near"], samples["far"], 400, 400).c
    for f in ("alpha_width", "alpha_height", "near", "far"):
        assert getattr(a, f) == getattr(b, f)
    for f in ("pos", "dir", "up"):
        assert np.allclose(list(getattr(a, f)), list(getattr(b, f)), atol=1e-7)


def test_packed_blob_round_trip(native, tmp_path):
    """nerf_pack_network_dir writes exactly the stream the directory loader would upload (host-only part)."""
    blob = tmp_path / "coarse.nrf"
    native.pack_network_dir(os.path.join(SCENE, "coarse"), blob)
    raw = blob.read_bytes()
    ws, sm = _pack(native, "coarse")
    assert raw[:8] == b"NRFMI355" and np.frombuffer(raw[8:16], np.uint32).tolist() == [1, ws.size + sm.size]
    assert np.array_equal(np.frombuffer(raw[16:], np.float32), np.concatenate([ws, sm]))
    with pytest.raises(native.NerfError):
        native.pack_network_dir(tmp_path / "missing", blob)


def test_bf16x3_split_is_exact_to_2pow27_and_six_products_match_f32(native):
    """The arithmetic behind NERF_MLP_BF16X3, on the host implementation of the split (the packer's): a value
    is the sum of its three bf16 parts up to 2^-27, and the six significant part products reproduce an f32 product to a
    fraction of an f32 ulp (checked in float64 on random weights x activations, including tiny and large magnitudes)."""
    import ctypes as C
    rng = np.random.default_rng(5)
    v = np.concatenate([rng.normal(size=20000), rng.normal(size=2000) * 1e-4, rng.normal(size=2000) * 1e3,
                        [0.0, 1.0, -1.0, 3.0e-39, 65504.0, 1.0 + 2.0 ** -23]]).astype(np.float32)
    parts = np.empty((v.size, 3), np.uint16)
    L = native.load_library()
    assert L.nerf_debug_split_bf16x3(v.ctypes.data_as(C.POINTER(C.c_float)), v.size, parts.ctypes.data_as(C.POINTER(C.c_uint16))) == 0
    p64 = (parts.astype(np.uint32) << 16).view(np.float32).astype(np.float64)     # bf16 bit pattern -> value
    resid = np.abs(v.astype(np.float64) - p64.sum(axis=1))
    assert (resid <= 2.0 ** -26 * np.abs(v) + 2.0 ** -132).all()                    # (+ the bf16 subnormal spacing for f32 subnormals)
    assert (np.abs(p64[:, 1]) <= 2.0 ** -8 * np.abs(p64[:, 0]) + 1e-44).all()       # parts shrink by at least 2^-8 each
    assert (np.abs(p64[:, 2]) <= 2.0 ** -8 * np.abs(p64[:, 1]) + 1e-44).all()
    w, x = p64[:12000], p64[12000:24000]                                             # pair weights with activations
    six = (w[:, 2] * x[:, 0] + w[:, 1] * x[:, 1] + w[:, 0] * x[:, 2]) + (w[:, 1] * x[:, 0] + w[:, 0] * x[:, 1]) + w[:, 0] * x[:, 0]
    exact = v[:12000].astype(np.float64) * v[12000:24000].astype(np.float64)
    assert (np.abs(six - exact) <= 2.0 ** -24 * np.abs(exact) + 1e-40).all()          # within half an f32 ulp of the true product


def test_f16x2_split_is_exact_enough(native):
    """The NERF_MLP_F16X2 packer's two-way f16 split: round to nearest even (numpy's float16 is the oracle), v = p0 + p1 to
    2^-22 |v| (+ the f16 subnormal spacing), and the three products the kernel forms approximate the f32 product to 2^-21."""
    rng = np.random.default_rng(5)
    v = np.concatenate([rng.normal(size=20000) * 0.13, rng.uniform(-8.3, 8.3, 2000), rng.normal(size=2000) * 300.0,
                        rng.normal(size=2000) * 1e-4, [0.0, 1.0, -1.0, 65504.0, 6.1e-5, 5.9e-8]]).astype(np.float32)
    parts = np.empty((v.size, 2), np.uint16)
    L = native.load_library()
    assert L.nerf_debug_split_f16x2(v.ctypes.data_as(C.POINTER(C.c_float)), v.size, parts.ctypes.data_as(C.POINTER(C.c_uint16))) == 0
    h = v.astype(np.float16)
    l = (v - h.astype(np.float32)).astype(np.float16)
    assert np.array_equal(parts[:, 0], h.view(np.uint16)) and np.array_equal(parts[:, 1], l.view(np.uint16))
    p64 = parts.view(np.float16).astype(np.float64)
    resid = np.abs(v.astype(np.float64) - p64.sum(axis=1))
    assert (resid <= 2.0 ** -22 * np.abs(v) + 2.0 ** -25).all()
    w, x = p64[:12000], p64[12000:24000]
    three = (w[:, 1] * x[:, 0] + w[:, 0] * x[:, 1]) + w[:, 0] * x[:, 0]
    exact = v[:12000].astype(np.float64) * v[12000:24000].astype(np.float64)
    assert (np.abs(three - exact) <= 2.0 ** -21 * np.abs(exact) + 2.0 ** -24 * (np.abs(v[:12000]) + np.abs(v[12000:24000])) + 1e-12).all()


def _c_prototypes(text):
    """name -> number of parameters, for every function declared in a C header (comments stripped)."""
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for name, args in re.findall(r"\b(nerf_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        args = args.strip()
        out[name] = 0 if args in ("", "void") else len(args.split(","))
    return out


def test_rust_sys_crate_declares_the_whole_header():
    """bindings/rust/nerf-mi355x-sys cannot be compiled here (no rustc in the image), so its `extern "C"` block is checked
    textually against include/nerf_mi355x.h: same function names, same number of arguments; and the #[repr(C)] struct mirrors
    list the same number of fields as the ctypes mirrors (whose sizes the library verifies at load time)."""
    from nerf_rs_amd import _lib
    header = _c_prototypes(open(HEADER).read())
    rs = open(os.path.join(ROOT, "bindings", "rust", "nerf-mi355x-sys", "src", "lib.rs")).read()
    rs_nc = re.sub(r"//.*", "", rs)
    block = rs_nc[rs_nc.index('extern "C" {'):]
    block = block[:block.index("\n}")]
    rust = {}
    for name, args in re.findall(r"pub fn (nerf_[a-z0-9_]+)\s*\(([^)]*)\)", block, flags=re.S):
        rust[name] = len([a for a in args.split(",") if a.strip()])
    assert set(rust) == set(header), (sorted(set(header) - set(rust)), sorted(set(rust) - set(header)))
    assert rust == header
    assert set(header) == set(_lib.PROTOTYPES) and all(len(_lib.PROTOTYPES[n][1]) == header[n] for n in header)
    for struct, mirror in (("nerf_camera", _lib.CCamera), ("nerf_render_opts", _lib.COpts), ("nerf_stats", _lib.CStats)):
        body = rs_nc[rs_nc.index(f"pub struct {struct} {{"):]
        body = body[:body.index("}")]
        fields = re.findall(r"pub ([a-z_0-9]+):", body)
        assert fields == [f[0] for f in mirror._fields_], (struct, fields)
    # the same functions in INTEGRATION.md's extern block (what a maintainer of the reference would paste)
    integ = re.sub(r"//.*", "", open(os.path.join(ROOT, "INTEGRATION.md")).read())
    integ_fns = set(re.findall(r"pub fn (nerf_[a-z0-9_]+)\s*\(", integ))
    assert integ_fns <= set(header) and {"nerf_create", "nerf_render_image", "nerf_render_image_multi", "nerf_forward_batch"} <= integ_fns


# ---- C declaration -> Rust FFI type, for the textual ABI check below (no Rust compiler exists in this image) ----------------
_C_SCALARS = {"int": "c_int", "int32_t": "i32", "uint32_t": "u32", "int64_t": "i64", "uint64_t": "u64", "uint16_t": "u16",
              "uint8_t": "u8", "size_t": "usize", "float": "f32", "double": "f64", "char": "c_char", "void": "c_void",
              "nerf_ctx": "nerf_ctx", "nerf_camera": "nerf_camera", "nerf_render_opts": "nerf_render_opts", "nerf_stats": "nerf_stats"}


def _c_decl_to_rust(decl):
    """`const float *const *data` -> ('data', '*const *const f32'); `float pos[3]` -> ('pos', '[f32; 3]') for struct fields and
    ('pos', '*const f32' / '*mut f32') for parameters is handled by the caller through `array`."""
    decl = decl.strip()
    m = re.match(r"^(.*?)([A-Za-z_][A-Za-z_0-9]*)\s*(\[(\d+)\])?$", decl)
    assert m, decl
    ty, name, _, arr = m.group(1).strip(), m.group(2), m.group(3), m.group(4)
    toks = re.findall(r"[A-Za-z_][A-Za-z_0-9]*|\*", ty)
    # C reads right to left: base [const] then a list of ('*', const-after?)
    base_const = False
    base = None
    ptrs = []
    for t in toks:
        if t == "const":
            if ptrs:
                ptrs[-1] = True          # `*const`: the pointer just seen is itself const
            else:
                base_const = True
        elif t == "*":
            ptrs.append(False)
        else:
            assert base is None, decl
            base = t
    rust = _C_SCALARS[base]
    # innermost pointer takes its mutability from the pointee's constness, each outer one from the const-ness of the pointer it points to
    pointee_const = base_const
    for self_const in ptrs:
        rust = ("*const " if pointee_const else "*mut ") + rust
        pointee_const = self_const
    return name, rust, (int(arr) if arr else None), base_const


def _c_struct_fields(text, struct):
    """[(name, rust type)] of `typedef struct { ... } struct;` in declaration order."""
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    body = re.search(r"typedef struct \{([^}]*)\}\s*" + struct + r"\s*;", text, flags=re.S).group(1)
    out = []
    for stmt in body.split(";"):
        stmt = " ".join(stmt.split())
        if not stmt:
            continue
        first, *rest = [x.strip() for x in stmt.split(",")]
        name, rust, arr, _ = _c_decl_to_rust(first)
        ty = first[:first.rindex(name)]
        for decl in [first] + [ty + r for r in rest]:
            name, rust, arr, _ = _c_decl_to_rust(decl)
            out.append((name, f"[{rust}; {arr}]" if arr else rust))
    return out


def _c_param_types(text):
    """function name -> [rust parameter types] for every prototype of the header."""
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for ret, name, args in re.findall(r"\b(int|void|const char \*)\s*(nerf_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        args = " ".join(args.split())
        types = []
        if args not in ("", "void"):
            for a in args.split(","):
                _, rust, arr, base_const = _c_decl_to_rust(a)
                types.append((("*const " if base_const else "*mut ") + rust) if arr else rust)   # `const float c2w[12]` decays to a pointer
        out[name] = (types, {"int": "c_int", "void": "", "const char *": "*const c_char"}[ret])
    return out


def test_rust_sys_crate_matches_the_header_types():
    """The `-sys` crate will never meet a compiler here, so field ORDER and TYPES of the #[repr(C)] mirrors, and the type of every
    argument and return value of the extern block, are derived from include/nerf_mi355x.h and compared textually."""
    from nerf_rs_amd import _lib
    htext = open(HEADER).read()
    rs = re.sub(r"//.*", "", open(os.path.join(ROOT, "bindings", "rust", "nerf-mi355x-sys", "src", "lib.rs")).read())
    ctypes_names = {C.c_int32: "i32", C.c_uint32: "u32", C.c_uint64: "u64", C.c_float: "f32", C.c_double: "f64", C.c_float * 3: "[f32; 3]",
                    C.c_float * 2: "[f32; 2]"}
    for struct, mirror in (("nerf_camera", _lib.CCamera), ("nerf_render_opts", _lib.COpts), ("nerf_stats", _lib.CStats)):
        want = _c_struct_fields(htext, struct)
        body = rs[rs.index(f"pub struct {struct} {{"):]
        body = body[:body.index("}")]
        got = [(n.rstrip("_"), " ".join(t.split())) for n, t in re.findall(r"pub ([a-z_0-9]+):\s*([^,\n]+),", body)]
        assert got == [(n.rstrip("_"), t) for n, t in want], (struct, got, want)          # `near_` / `far_` are `near` / `far` in Rust
        py = [(n.rstrip("_"), ctypes_names[t]) for n, t in mirror._fields_]
        assert py == [(n.rstrip("_"), t) for n, t in want], (struct, py, want)             # and the ctypes mirror agrees as well
    block = rs[rs.index('extern "C" {'):]
    block = block[:block.index("\n}")]
    want = _c_param_types(htext)
    got = {}
    for name, args, ret in re.findall(r"pub fn (nerf_[a-z0-9_]+)\s*\(([^)]*)\)\s*(->\s*[^;]+)?;", block, flags=re.S):
        types = [" ".join(a.split(":", 1)[1].split()) for a in args.split(",") if a.strip()]
        got[name] = (types, " ".join(ret.replace("->", "").split()) if ret else "")
    assert set(got) == set(want)
    for name in want:
        assert got[name] == want[name], (name, got[name], want[name])
    consts = dict(re.findall(r"pub const (NERF_[A-Z0-9_]+): (?:c_int|i32) = (-?\d+);", rs))
    henum = dict(re.findall(r"\b(NERF_[A-Z0-9_]+)\s*=\s*(-?\d+)", re.sub(r"/\*.*?\*/", "", htext, flags=re.S)))
    assert consts == henum, (sorted(set(henum) ^ set(consts)), {k: (consts.get(k), henum.get(k)) for k in henum if consts.get(k) != henum.get(k)})


def test_struct_sizes_match_the_library(native):
    from nerf_rs_amd import _lib
    L = native.load_library()
    a, b, c = C.c_size_t(), C.c_size_t(), C.c_size_t()
    L.nerf_abi_struct_sizes(C.byref(a), C.byref(b), C.byref(c))
    assert (a.value, b.value, c.value) == (C.sizeof(_lib.CCamera), C.sizeof(_lib.COpts), C.sizeof(_lib.CStats)) == (60, 72, 160)


def test_loader_rejects_a_directory_named_like_a_tensor(native, tmp_path):
    """A <name>.bin that is a directory (ftell gives -1 or nonsense) must be an I/O error, not a giant allocation or an
    exception across the C ABI."""
    d = tmp_path / "net"; shutil.copytree(os.path.join(SCENE, "coarse"), d)
    os.remove(d / "dense2_bias.bin"); os.mkdir(d / "dense2_bias.bin")
    rc, msg = _check(native, d)
    assert rc == -2 and "read tensor" in msg and "dense2_bias" in msg


def test_render_opts_mirror_maps_every_field(native):
    """The Python mirror of nerf_render_opts: every option reaches its C field (no device needed)."""
    from nerf_rs_amd.api import RenderOpts, _DTYPES
    assert (_DTYPES["f32"], _DTYPES["bf16"], _DTYPES["bf16x3"], _DTYPES["f16x2"]) == (0, 1, 2, 3)
    o = RenderOpts(n_coarse=40, n_fine=50, coarse_only=True, crop=(1, 2, 3, 4), ssaa=2, seed=(1 << 40) + 7, dtype="f16x2",
                   skip_empty=True, skip_dead=True, hybrid_sampling=True, certify_zero=True, band=(2, 5, 1)).to_c()
    got = {name: getattr(o, name) for name, _ in type(o)._fields_}
    assert got == {"n_coarse": 40, "n_fine": 50, "coarse_only": 1, "crop_x0": 1, "crop_y0": 2, "crop_w": 3, "crop_h": 4, "ssaa": 2,
                   "seed": (1 << 40) + 7, "mlp_dtype": 3, "skip_empty": 1, "skip_dead": 1, "hybrid_sampling": 1, "certify_zero": 1,
                   "band_index": 2, "band_count": 5, "band_stripe_rows": 1}
    z = RenderOpts().to_c()
    assert (z.n_coarse, z.n_fine, z.mlp_dtype, z.skip_empty, z.skip_dead, z.hybrid_sampling, z.crop_w, z.ssaa, z.band_count) == (64, 128, 0, 0, 0, 0, 0, 1, 0)


def test_band_partitions_cover_the_window_once(native):
    """nerf_band_rows (the C side) and band_row_indices (the layout the bands are packed in) agree, and every partition -- contiguous,
    stripes of 1, 3, 8 rows -- deals every row of the window out exactly once, in frame order within a band."""
    for h in (1, 7, 8, 61, 100, 800, 801):
        for n in (1, 2, 3, 4, 8, 11):
            for stripe in (0, 1, 3, 8):
                rows = [native.band_row_indices(h, i, n, stripe) for i in range(n)]
                assert [len(r) for r in rows] == [native.band_rows(h, i, n, stripe) for i in range(n)]
                assert sorted(np.concatenate(rows).tolist()) == list(range(h))
                assert all((np.diff(r) > 0).all() for r in rows)
                assert len(rows[0]) == max(len(r) for r in rows)     # equal-slot gathers size their slots by band 0
                if stripe == 1 or stripe == 0:
                    assert max(len(r) for r in rows) - min(len(r) for r in rows) <= 1
    assert native.load_library().nerf_band_rows(10, 3, 3, 0) < 0 and native.load_library().nerf_band_rows(-1, 0, 1, 0) < 0


def test_loader_refuses_a_variant_build(native, tmp_path):
    """`make variant` libraries (tuning switches, timing-only diagnostics that make results WRONG on purpose) report a build tag;
    the loader only accepts them with NERF_ALLOW_VARIANT=1, so NERF_MI355X_LIB cannot silently replace the product library."""
    import subprocess
    import sys
    src = tmp_path / "fake.c"
    names = "\n".join(f"void {n}(void) {{}}" for n in native._lib.PROTOTYPES if n != "nerf_build_variant")
    src.write_text(names + '\nconst char *nerf_build_variant(void) { return "probe: -DNERF_DIAG_NO_DMA=1"; }\n')
    so = tmp_path / "libfake.so"
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-o", str(so), str(src)])
    code = "import nerf_rs_amd; nerf_rs_amd.load_library()"
    env = dict(os.environ, NERF_MI355X_LIB=str(so), PYTHONPATH=ROOT)
    env.pop("NERF_ALLOW_VARIANT", None)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert p.returncode != 0 and "tuning variant (probe: -DNERF_DIAG_NO_DMA=1)" in p.stderr


def test_make_variant_tags_every_compile_line():
    """The Makefile's own variant recipe (not a hand-made stand-in): with DEFS on the command line -- the documented invocation -- every
    compile line must carry -DNERF_BUILD_VARIANT, or nerf_build_variant() returns "" and the loader takes a timing-only build for the
    product (round 3: a target-specific `DEFS +=` was ignored for command-line variables)."""
    import subprocess
    csrc = os.path.join(ROOT, "nerf-rs_amd", "csrc")
    out = subprocess.run(["make", "-n", "-C", csrc, "variant", "NAME=x", "DEFS=-DA=1"], capture_output=True, text=True, check=True).stdout
    compiles = [l for l in out.splitlines() if " -c " in l]
    assert len(compiles) >= 11
    for l in compiles:
        assert "-DA=1" in l and "-DNERF_BUILD_VARIANT='\"x: -DA=1\"'" in l, l
    bare = subprocess.run(["make", "-n", "-C", csrc, "variant", "NAME=y"], capture_output=True, text=True, check=True).stdout
    assert all("-DNERF_BUILD_VARIANT='\"y: \"'" in l for l in bare.splitlines() if " -c " in l)
    assert subprocess.run(["make", "-C", csrc, "variant"], capture_output=True, text=True).returncode != 0  # NAME is required


def test_abi_version_is_stated_consistently(native):
    """One number, five places: nerf_abi_version(), the header's history comment, the Rust -sys crate (doc line and check_layouts),
    DESIGN.md's boundary row and INTEGRATION.md (round 3 left three of them behind)."""
    v = native.load_library().nerf_abi_version()
    h = open(HEADER).read()
    assert f"ABI version (currently {v})" in h and re.search(rf"\b{v}: nerf_", h)
    assert not re.search(rf"\b{v + 1}: nerf_", h)
    rs = open(os.path.join(ROOT, "bindings", "rust", "nerf-mi355x-sys", "src", "lib.rs")).read()
    assert f"(ABI version {v})" in rs.splitlines()[0] and f"nerf_abi_version() }} == {v} " in rs and f"expects ABI {v} with" in rs
    assert re.search(rf"C ABI \(ABI version {v}[):]", open(os.path.join(ROOT, "DESIGN.md")).read())
    assert f"`nerf_abi_version` ({v})" in open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert f"nerf_abi_version() == {v}" in open(os.path.join(ROOT, "__graft_entry__.py")).read()


def test_certify_audit_policy(native):
    """certify_zero's audit policy (host_util.cpp certify_policy, the function render_device applies after every certified frame), on the CPU:
    a frame stands only if no audited certificate was wrong, the closest audited sample kept half the margin, and the bf16 pass was off by
    at most half the margin on every audited certificate; every rule widens the margin enough that the same audit would pass afterwards."""
    L = native.load_library()
    inf = float("inf")

    def policy(m, audited, viol, head, err):
        out = C.c_float(-1.0)
        return L.nerf_debug_certify_policy(m, audited, viol, head, err, C.byref(out)), out.value

    assert policy(3.0, 1_700_000, 0, 2.39, 0.99) == (0, 3.0)                 # the lego fine network, C3 frame
    assert policy(1.0, 50_000, 0, 0.96, 0.15) == (0, 1.0)                    # the coarse one
    assert policy(3.0, 0, 0, inf, 0.0) == (0, 3.0)                           # nothing certified (a random-weight fog): nothing to judge
    rule, m = policy(3.0, 1000, 2, 1.0, 29.0)                                # a wrong certificate: x 4, or 4 x the error seen
    assert rule == 1 and m == pytest.approx(116.0)
    assert policy(3.0, 1000, 1, 2.9, 0.2) == (1, 12.0)
    rule, m = policy(3.0, 1000, 0, 1.2, 0.3)                                 # headroom 1.2 < 1.5: boundary error 1.8 -> 4 x 1.8
    assert rule == 2 and m == pytest.approx(7.2)
    rule, m = policy(3.0, 1000, 0, 2.9, 1.6)                                 # off by 1.6 > 1.5 somewhere below the margin -> 3 x 1.6
    assert rule == 3 and m == pytest.approx(4.8)
    assert policy(3.0, 1000, 0, 2.9, 1.5)[0] == 0                            # exactly half the margin still stands
    assert policy(3.0, 1000, 0, 2.9, float("nan"))[0] == 3                   # a NaN statistic never lets a frame stand
    # after a widening the same audit passes, and a hot network converges: errors scale with the network, margins follow
    rng = np.random.default_rng(3)
    for _ in range(200):
        m0 = float(rng.uniform(0.5, 5)); err = float(rng.uniform(0, 50)); head = float(rng.uniform(-1, 1) * m0 + m0 * 0.5)
        viol = int(rng.integers(0, 2)) if head < m0 else 0
        rule, m1 = policy(m0, 1000, viol, head, err)
        if rule:
            assert m1 >= 1.25 * m0 and policy(m1, 1000, 0, m1 - min(err, m0 - head if head < m0 else 0.0), err)[0] in (0, 3)
            assert policy(m1, 1000, 0, m1, min(err, 0.5 * m1))[0] == 0
    assert L.nerf_debug_certify_policy(0.0, 1, 0, 1.0, 0.0, None) < 0


def test_one_file_variant_script_tags_its_library():
    """tools/variant_one.sh (one kernel file rebuilt with tuning switches) must tag its library like `make variant` does: until round 4 it did not."""
    src = open(os.path.join(ROOT, "tools", "variant_one.sh")).read()
    assert "-DNERF_BUILD_VARIANT=" in src and "nerf_host_api.cpp" in src and "amdgpu-mfma-vgpr-form=1" in src


def test_m0_belongs_to_the_lds_dma_statements(native, tmp_path):
    """The f32 and bf16 MLP kernels write M0 -- the LDS destination of their LDS-DMA pieces -- once per chunk quarter and leave it alone until
    the next chunk (mlp_common.hip.h dma_set_dst / glds_piece_m0), which is only right if hipcc uses M0 for nothing of its own.  Checked on
    the ISA of the library as built: every instruction that names m0 is one of the kernels' own s_mov_b32, no instruction with an implicit
    M0 operand exists, and where M0 is never read back (the per-chunk kernels) four pieces follow every write."""
    bundler, objdump, objcopy = (f"/opt/rocm/lib/llvm/bin/{t}" for t in ("clang-offload-bundler", "llvm-objdump", "llvm-objcopy"))
    if not all(os.path.exists(t) for t in (bundler, objdump, objcopy)):
        pytest.skip("ROCm llvm tools not installed")
    import subprocess
    fat = tmp_path / "fat.bin"
    subprocess.run([objcopy, "--dump-section", f".hip_fatbin={fat}", native.lib_path(), str(tmp_path / "ignored.so")], check=True)
    data = fat.read_bytes()
    starts = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data)]
    assert len(starts) >= 5  # one bundle per kernel file
    per_chunk_kernels = 0
    for i, s in enumerate(starts):
        e = starts[i + 1] if i + 1 < len(starts) else len(data)
        b, co = tmp_path / f"b{i}.bin", tmp_path / f"b{i}.co"
        b.write_bytes(data[s:e])
        subprocess.run([bundler, "--unbundle", "--type=o", f"--input={b}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        isa = subprocess.run([objdump, "-d", str(co)], check=True, capture_output=True, text=True).stdout
        writes = reads = pieces = 0
        for line in isa.splitlines():
            ins = line.split("//")[0].strip()
            assert not re.match(r"(s_movrel|v_movrel|s_sendmsg|ds_gws|s_set_gpr_idx|v_interp|ds_\w*addtid|\w+ .*lds_direct)", ins), ins
            if "global_load_lds_dwordx4" in ins:
                pieces += 1
            if re.search(r"\bm0\b", ins):
                if re.fullmatch(r"s_mov_b32 m0, s\d+", ins):
                    writes += 1
                elif re.fullmatch(r"s_mov_b32 s\d+, m0", ins):
                    reads += 1
                else:
                    raise AssertionError(f"an instruction of hipcc's own touches m0: {ins}")
        if writes and not reads:  # M0 per chunk quarter
            assert pieces == 4 * writes, (i, pieces, writes)
            per_chunk_kernels += 1
        elif writes:              # saved, written, restored around every piece (the split arithmetics)
            assert writes == 2 * reads and pieces == reads, (i, pieces, writes, reads)
    assert per_chunk_kernels >= 3  # mlp_kernel, mlp_kernel_seq, mlp_kernel_bf16v2


def test_header_is_plain_c_and_links(native, tmp_path):
    """The boundary is a C ABI: include/nerf_mi355x.h must compile as strict C99 (what a cgo / Rust bindgen / plain C host sees) and a C
    program must link against the library and call its host-only entry points (no GPU needed for these)."""
    if not shutil.which("gcc"):
        pytest.skip("gcc not installed")
    import subprocess
    src = tmp_path / "host.c"
    src.write_text('#include <stdio.h>\n#include "nerf_mi355x.h"\n'
                   'int main(void) {\n'
                   '    nerf_render_opts o; nerf_camera cam; nerf_stats st;\n'
                   '    (void)o; (void)cam; (void)st;\n'
                   '    printf("%d %s\\n", (int)nerf_abi_version(), nerf_build_variant());\n'
                   '    return 0;\n}\n')
    exe = tmp_path / "host"
    libdir = os.path.dirname(native.lib_path())
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", f"-I{os.path.join(ROOT, 'include')}", str(src), "-o", str(exe),
                    f"-L{libdir}", "-lnerf_mi355x", f"-Wl,-rpath,{libdir}"], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert int(out[0]) == native.abi_version() if hasattr(native, "abi_version") else int(out[0]) >= 5
