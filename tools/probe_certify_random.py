import os, sys, pathlib, tempfile
import numpy as np
ROOT='/root/repo' if os.path.exists('/root/repo/tools') else os.environ.get('GRAFT_REPO_ROOT','.')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tools'))
import nerf_rs_amd as N
from scene_utils import random_scene
import json
S=json.load(open(os.path.join(ROOT,'lego_rust','tf_reference_samples.json')))
tmp=pathlib.Path(tempfile.mkdtemp())
for seed,kw in ((321,{}),(99,{}),(7,dict(alpha_bias=(2.0,2.0),alpha_scale=0.01))):
    root=random_scene(tmp/f'rnd{seed}',seed,**kw)
    with N.Renderer(0) as r:
        co=N.load_network_from_dir(r,0,root/'coarse'); fi=N.load_network_from_dir(r,1,root/'fine')
        cam=N.camera_from_samples(S,128,128,64)
        a=N.render_image(co,fi,cam,128,seed=5)
        b,st=N.render_image(co,fi,cam,128,seed=5,certify_zero=True,return_stats=True)
        d=np.abs(a-b)
        print('random scene',seed,kw,'identical',np.array_equal(a,b),'max %.2e'%d.max(),'pixels differing',int((d.max(axis=2)>0).sum()),'f32 evaluates coarse %.3f fine %.3f'%(st.n_exec_coarse_trunk/st.n_coarse_points, st.n_exec_fine_trunk/st.n_fine_points))
