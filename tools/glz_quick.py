import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spice_amd
from spice_amd import binding as B, frames
import ctypes
gpu=B.SpiceGpu(ctypes.CDLL(os.environ['GLZ_LIB'])) if os.environ.get('GLZ_LIB') else None
if gpu: gpu.init(0)
else: gpu=spice_amd.load(0)
S=int(sys.argv[1]) if len(sys.argv)>1 else 16
NF=int(sys.argv[2]) if len(sys.argv)>2 else 6
W,H=1920,1080
t0=time.time()
seqs=[frames.DesktopSequence(W,H,seed=100+s) for s in range(min(S,int(os.environ.get('POOL','8'))))]
fr=[[seq.next() for _ in range(NF)] for seq in seqs]
print('gen',time.time()-t0,flush=True)
pool=[[gpu.to_device(fr[p][k]) for k in range(NF)] for p in range(len(fr))]
dev=[pool[s%len(fr)] for s in range(S)]
dicts=[B.GlzDictionary(gpu,16*1024*1024,1) for s in range(S)]
encs=[B.GlzEncoder(gpu,d,0) for d in dicts]
cap=gpu.L.spice_gpu_glz_out_bound(8,W,H)
outs=[gpu.malloc(cap) for s in range(S)]
for k in range(NF):
    items=[dict(encoder=encs[s],type=8,width=W,height=H,stride=W*4,chunks=[dev[s][k]],chunk_rows=[H],out=outs[s],out_capacity=cap) for s in range(S)]
    gpu.L.spice_gpu_timing_enable(1); gpu.L.spice_gpu_timing_reset()
    t=time.time()
    sizes,_=B.glz_encode_batch(gpu,items)
    dt=time.time()-t
    kt=gpu.kernel_times()
    print('frame',k,'ms %.2f'%(dt*1e3),'Mpx/s %.0f'%(S*W*H/dt/1e6),'out avg',int(np.mean(sizes)),'parse ms %.2f copy ms %.2f'%(kt.get('glz_parse_kernel',(0,0))[1],kt.get('glz_copy_kernel',(0,0))[1]),flush=True)
