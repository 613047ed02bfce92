import sys, os, numpy as np
sys.path.insert(0,'tests'); sys.path.insert(0,'oracle'); sys.path.insert(0,'.')
import importlib.util, conftest
spec=importlib.util.spec_from_file_location('tg','tests/test_gpu_parity.py'); tg=importlib.util.module_from_spec(spec); spec.loader.exec_module(tg)
import nerf_rs_amd as N
g=np.load('tests/golden/forward_batch_4096.npz')
with N.Renderer(0) as r:
    r.load_scene('lego_rust')
    for sub,net in (('coarse',r.coarse),('fine',r.fine)):
        rgb64,sg64=tg._forward_fp64(sub,g['pts'],g['dirs'])
        rgb,sg=net.forward_batch(g['pts'],g['dirs'])
        print(sub,'GPU f32 vs fp64: sigma rel max %.2e rgb max %.2e'%((np.abs(sg-sg64)/(1+np.abs(sg64))).max(), np.abs(rgb-rgb64).max()))
