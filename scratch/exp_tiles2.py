import sys; sys.path.insert(0,'.')
import numpy as np, torch
import xmipp3_amd as xa
from tests import synth
ctx=xa.Context(0)
D=256; B=64
rf=xa.RecFourier(ctx,D)
g=torch.Generator(device='cuda'); g.manual_seed(1)
imgs=torch.randn((B,D,D),generator=g,device='cuda')
fft=rf.prepare_images(imgs)
ang=synth.random_angles(B,np.random.default_rng(0))
for dbg in (0,2):
    rf.set_option("tile_dbg",dbg)
    for _ in range(3):
        rf.insert(fft,ang)
    ctx.sync()
