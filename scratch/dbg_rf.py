import sys; sys.path.insert(0,'.')
import numpy as np, torch
import xmipp3_amd as xa
from oracle import pyoracle as o
ctx=xa.Context(0)
for D in (32,64,128):
    rng=np.random.default_rng(D)
    imgs=rng.standard_normal((3,D,D)).astype(np.float32)
    rf=xa.RecFourier(ctx,D); orf=o.RF(D)
    got=rf.prepare_images(torch.from_numpy(imgs).cuda()).cpu().numpy()
    for i in range(3):
        exp=orf.prepare_image(imgs[i])
        err=np.abs(got[i]-exp).max(axis=(1,2))
        bad=np.nonzero(err>1e-5*np.abs(exp).max())[0]
        print(D,i,"max err",err.max(),"peak",np.abs(exp).max(),"bad rows",len(bad), bad[:10], bad[-5:])
        errc=np.abs(got[i]-exp).max(axis=(0,2)); badc=np.nonzero(errc>1e-5*np.abs(exp).max())[0]
        print("   bad cols",len(badc),badc[:10])
