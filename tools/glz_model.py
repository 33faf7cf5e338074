import sys, ctypes, numpy as np
import os; ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from oracle import oracle as O
from spice_amd import frames
W,H=1920,1080
L=O.lib()
L.glzo_set_trace.argtypes=[ctypes.c_void_p, ctypes.c_size_t]
L.glzo_trace_count.restype=ctypes.c_size_t
seed=int(sys.argv[1]) if len(sys.argv)>1 else 100
WIN=int(sys.argv[2]) if len(sys.argv)>2 else 256
seq=frames.DesktopSequence(W,H,seed=seed)
d=O.GlzDict(16*1024*1024,1)
buf=np.zeros(2*4_000_000,dtype=np.uint32)
def hashes(px):
    p=px&0xffffff
    a=p[:-2];b=p[1:-1];c=p[2:]
    # any injective-ish combination is fine for classification: use exact triple identity via hashing to 64-bit then to djb-like 20 bit is what matters for collisions; approximate with real djb
    h=np.full(a.shape,5381,dtype=np.uint32)
    for v in (a,b,c):
        for sh in (16,8,0):
            h=((h<<np.uint32(5))+h)^((v>>np.uint32(sh))&np.uint32(0xff))
    return h&np.uint32((1<<20)-1)
for k in range(3):
    f=seq.next().reshape(H,-1)
    L.glzo_set_trace(buf.ctypes.data, 4_000_000)
    d.encode(O.TYPE_RGB32,W,H,W*4,[f])
    n=L.glzo_trace_count()
    t=buf[:2*n].reshape(-1,2).copy()
    pos=t[:,0].astype(np.int64); raw=t[:,1]; ln=(raw&0x3fffffff).astype(np.int64)
    rle=(raw>>31)&1; other=(raw>>30)&1
    px=f.view(np.uint32).reshape(-1)
    hh=hashes(px)
    N=px.size
    inserted=np.zeros(N+4,dtype=bool)
    cls={'rle':0,'noprev':0,'p1':0,'p2':0,'deeper_ins':0,'none_ins':0}
    mlen_hist=np.zeros(8,dtype=np.int64)
    long_inwin=0; long_glob=0
    i=0
    while i<n:
        start=pos[i]
        # per window: build last-occurrence chains lazily
        while i<n and pos[i]<start+WIN:
            q=pos[i]
            if rle[i]:
                cls['rle']+=1
            else:
                h=hh[q]
                # predecessors with same hash in [start,q)
                seg=hh[start:q]
                idx=np.nonzero(seg==h)[0]
                if idx.size==0:
                    cls['noprev']+=1
                    if ln[i]>16: long_glob+=1
                else:
                    ins=inserted[start+idx]
                    if ins[-1]: cls['p1']+=1
                    elif idx.size>1 and ins[-2]: cls['p2']+=1
                    elif ins.any(): cls['deeper_ins']+=1
                    else: cls['none_ins']+=1
                    if ln[i]>16 and ins.any(): long_inwin+=1
                inserted[q]=True
            if ln[i]>0:
                p0=q+ln[i]-2
                if p0>q: inserted[p0]=True
                if p0+1>q: inserted[p0+1]=True
                b=min(7,int(ln[i]).bit_length())
                mlen_hist[b]+=1
            i+=1
    print('frame',k,'stops',n,cls,'len>16 inwin',long_inwin,'glob',long_glob,'lenhist(bitlen)',mlen_hist.tolist(), 'len>8',int((ln>8).sum()),'len>16',int((ln>16).sum()),'len>32',int((ln>32).sum()), 'otherimg',int(other.sum()))
