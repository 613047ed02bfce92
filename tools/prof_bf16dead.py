import os, sys
sys.path.insert(0, "/root/repo")
import nerf_rs_amd as N
ROOT="/root/repo"
with N.Renderer(0) as r:
    r.load_scene(os.path.join(ROOT, "lego_rust"))
    cam = N.camera_from_samples(os.path.join(ROOT, "lego_rust", "tf_reference_samples.json"), 800, 800, 64)
    for k in range(4):
        N.render_image(r.coarse, r.fine, cam, 128, seed=0, dtype="bf16", skip_dead=True)
    for k in range(4):
        N.render_image(r.coarse, r.fine, cam, 128, seed=0, dtype="bf16")
