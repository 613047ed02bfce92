#!/usr/bin/env python3
"""CPU experiment (negative result): could certify_zero's pre-filter leave the trunk early?  A linear probe (ridge regression) from the hidden
activations after layer k to the density pre-activation, fitted on half of the samples of real lego rays and tested on the other half (on the f16 pass's
own activations): its error on true zeros, the most negative value it predicts for a LIVE sample, and what it could certify at 2 x / 3 x that margin.
Result: after layers 1..6 the probes are off by tens to hundreds (fine network, after layer 6: median 1.4, p99 12.7, max 39; a live sample predicted at
-51) and certify nothing; only the probe after layer 7 works -- that is the alpha head itself.  The density is decided in the last layers: no early exit.
No GPU: sample positions from the CPU oracle (1 500 random pixels of the 800 x 800 frame), as tools/emulate_prefilter_error.py."""
import numpy as np, sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,ROOT+'/oracle'); sys.path.insert(0,ROOT+'/tools')
import oracle_py as O
from emulate_prefilter_error import load, enc, f16
cam=O.camera_from_samples(O.load_samples(ROOT+'/lego_rust/tf_reference_samples.json'),800,800)
_co,_fi=O.Net(ROOT+'/lego_rust/coarse'),O.Net(ROOT+'/lego_rust/fine')
_opts=O.make_opts(64,128,seed=0); _rng=np.random.default_rng(1); _rows=[]
for _ in range(int(sys.argv[1]) if len(sys.argv)>1 else 1500):
    _d=O.render_ray_debug(_co,_fi,cam,_opts,int(_rng.integers(0,800)),int(_rng.integers(0,800)))
    _rows.append(np.concatenate([_d['t_merged'],_d['dir_hat'],_d['t_coarse']]))
rays=np.array(_rows,np.float32)   # [t_merged(192) | dir_hat(3) | t_coarse(64)] per ray
o=np.array(list(cam.pos),np.float32)
def hidden(W,pts,rnd):
    e=enc(pts,10); h=e; H=[]
    for i in range(8):
        x=np.concatenate([e,h],1) if i==5 else h
        h=np.maximum(rnd(x)@rnd(W[f'dense{i}_kernel'])+W[f'dense{i}_bias'],0).astype(np.float32)
        H.append(h)
    return H,(h@W['alpha_kernel']+W['alpha_bias'])[:,0]
for net,ts in (('fine',slice(0,192)),('coarse',slice(195,259))):
    t=rays[:,ts]; d=rays[:,192:195]
    nr=t.shape[0]; half=nr//2
    pts=(o[None,None,:]+d[:,None,:]*t[:,:,None]).astype(np.float32)
    tr=pts[:half].reshape(-1,3); te=pts[half:].reshape(-1,3)
    W=load(net)
    Htr,ytr=hidden(W,tr,lambda x:x); Hte,yte=hidden(W,te,f16)   # fit on f32 activations, test on the f16 pass's own activations
    _,yte_exact=hidden(W,te,lambda x:x)
    zeros=yte_exact<=0
    print(net,'train',len(ytr),'test',len(yte),'zeros',zeros.mean())
    for k in range(1,8):
        A=np.concatenate([Htr[k],np.ones((len(ytr),1),np.float32)],1).astype(np.float64)
        # ridge
        lam=1e-3*len(ytr)
        w=np.linalg.solve(A.T@A+lam*np.eye(A.shape[1]),A.T@ytr.astype(np.float64))
        pred=np.concatenate([Hte[k],np.ones((len(yte),1),np.float32)],1).astype(np.float64)@w
        err=pred-yte_exact
        # dangerous direction: pred more negative than exact (err<0) on samples; overall abs error on zeros
        ez=np.abs(err[zeros]); eall=np.abs(err)
        # margin such that no test sample with exact>0 has pred < -m : smallest safe margin on the test set
        live=yte_exact>0
        worst_live=(-pred[live]).max() if live.any() else 0   # a live sample predicted at -worst_live
        for m in (2*max(worst_live,0)+1e-9, 3*max(worst_live,0)+1e-9):
            cert=(pred<-m)
            print(f'  probe after layer {k}: |err| on zeros p50 {np.median(ez):.2f} p99 {np.quantile(ez,.99):.2f} max {ez.max():.2f}; most negative prediction of a LIVE sample {-worst_live:.2f}; margin {m:.2f}: certifies {cert.mean():.3f} of all samples ({(cert&zeros).sum()/zeros.sum():.3f} of the zeros)')
    # ray-level: fraction of 32-sample chunks all certified (proxy for wave/WG-uniform exit)
