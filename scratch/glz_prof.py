import sys, time, ctypes, numpy as np
sys.path.insert(0,'.')
from spice_amd import binding as B, frames
L=ctypes.CDLL('scratch/libspicegpu_prof.so')
gpu=B.SpiceGpu(L); gpu.init(0)
S=int(sys.argv[1]) if len(sys.argv)>1 else 1
NF=int(sys.argv[2]) if len(sys.argv)>2 else 4
seed=int(sys.argv[3]) if len(sys.argv)>3 else 100
W,H=1920,1080
seq=frames.DesktopSequence(W,H,seed=seed)
fr=[seq.next() for _ in range(NF)]
dev=[gpu.to_device(f) for f in fr]
dicts=[B.GlzDictionary(gpu,16*1024*1024,1) for s in range(S)]
encs=[B.GlzEncoder(gpu,d,0) for d in dicts]
cap=gpu.L.spice_gpu_glz_out_bound(8,W,H)
outs=[gpu.malloc(cap) for s in range(S)]
prof=(ctypes.c_ulonglong*32)()
names=['tile','scan','resolve','emit','windows','x5','qsetup_t','quarters','general','general_t','loop_t','qpub_t','x12','x13','x14','x15','sc_bar_t','sc_search_t','sc_tabissue_t','sc_all_t','em_stores_t','em_runs_t','em_tok_t','em_scan_t','em_write_t']
for k in range(NF):
    items=[dict(encoder=encs[s],type=8,width=W,height=H,stride=W*4,chunks=[dev[k]],chunk_rows=[H],out=outs[s],out_capacity=cap) for s in range(S)]
    L.spice_gpu_debug_prof(prof,1)
    t=time.time()
    sizes,_=B.glz_encode_batch(gpu,items)
    dt=time.time()-t
    L.spice_gpu_debug_prof(prof,1)
    v=list(prof)
    print('frame',k,'ms %.2f'%(dt*1e3), {n:(v[i]/100.0 if n in('tile','scan','resolve','emit') or n.endswith('_t') else v[i]) for i,n in enumerate(names)},flush=True)
