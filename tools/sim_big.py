import sys, os, ctypes, time, numpy as np
import os; ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
from spice_amd import binding as B, frames
from oracle import oracle as O
from helpers import pixels_for, GlzPair, split_rows
gpu = B.SpiceGpu(ctypes.CDLL(os.path.join(ROOT,'tests/sim/libspicegpu_sim.so'))); gpu.init(0)
W,H=int(sys.argv[1]),int(sys.argv[2]); NF=int(sys.argv[3])
types=[int(x) for x in sys.argv[4].split(',')] if len(sys.argv)>4 else (8,9,7,6)
rng=np.random.default_rng(5)
for typ in types:
    seq=frames.DesktopSequence(W,H,seed=31+typ)
    pair=GlzPair(gpu,O,W*H*3,1)
    for k in range(NF):
        f=seq.next().reshape(H,W,4)
        img=pixels_for(typ,f,rng)
        t=time.time()
        pair.encode(typ,W,H,img.shape[1],split_rows(img,1+k%3),what='big')
        print('type',typ,'frame',k,'ok %.1fs'%(time.time()-t),flush=True)
    pair.close()
