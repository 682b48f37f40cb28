import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
import xmipp3_amd as xa
from tests import synth
ctx=xa.Context(0)
D=256; B=1024
rf=xa.RecFourier(ctx,D)
g=torch.Generator(device='cuda'); g.manual_seed(1)
imgs=torch.randn((B,D,D),generator=g,device='cuda')
fft=rf.prepare_images(imgs)
ang=synth.random_angles(B,np.random.default_rng(0))
ctfa=torch.rand((B,rf.sizeY,rf.sizeX),device='cuda')+0.5; moda=torch.rand((B,rf.sizeY,rf.sizeX),device='cuda')
t=ctx.timer()
for name,kw in (("tiles-noctf",{}),("tiles-ctf",dict(ctf=ctfa,modulator=moda))):
    rf.reset(); rf.insert(fft,ang,**kw); ctx.sync()
    t.start(); rf.insert(fft,ang,**kw); t.stop(); ms=t.elapsed_ms()
    print(name,"ms",ms,"us/particle",1e3*ms/B)
for dbg in (1,2,3,4,0):
    rf.set_option("tile_dbg",dbg)
    rf.insert(fft,ang); ctx.sync(); t.start(); rf.insert(fft,ang); t.stop(); print("tile_dbg",dbg,"us/particle",1e3*t.elapsed_ms()/B)
for Bs in (64,256,512):
    rf.insert(fft[:Bs].contiguous(),ang[:Bs]); ctx.sync(); t.start(); rf.insert(fft[:Bs].contiguous(),ang[:Bs]); t.stop(); print("B",Bs,"us/particle",1e3*t.elapsed_ms()/Bs)
sys.exit(0)
rf.set_option("tile_min_spaces",1<<30)
for name,kw in (("noctf",{}),("ctf",dict(ctf=ctfa,modulator=moda))):
  for v in (0,1,2):
    rf.set_option("insert_variant",v)
    rf.insert(fft,ang,**kw); ctx.sync()
    t.start(); rf.insert(fft,ang,**kw); t.stop(); ms=t.elapsed_ms()
    print(name,"variant",v,"ms",ms,"us/particle",1e3*ms/B)
rf.set_option("insert_variant",0)
# sorted by direction? tilt near 90 vs random
for nm,a in (("tilt0",np.stack([np.linspace(0,360,B),np.full(B,1.0),np.zeros(B)],1)),("tilt90",np.stack([np.linspace(0,360,B),np.full(B,90.0),np.zeros(B)],1)),("rot90tilt90",np.stack([np.full(B,90.0)+np.linspace(0,1,B),np.full(B,90.0),np.zeros(B)],1))):
    rf.insert(fft,a); ctx.sync(); t.start(); rf.insert(fft,a); t.stop(); print(nm,"ms",t.elapsed_ms(),"us/particle",1e3*t.elapsed_ms()/B)
# prepare timing
t.start(); fft=rf.prepare_images(imgs); t.stop(); print("prepare us/particle",1e3*t.elapsed_ms()/B)
