import sys, ctypes, numpy as np
import os; ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from oracle import oracle as O
from spice_amd import frames
W,H=1920,1080
L=O.lib()
L.glzo_set_trace.argtypes=[ctypes.c_void_p, ctypes.c_size_t]
L.glzo_trace_count.restype=ctypes.c_size_t
seed=int(sys.argv[1]) if len(sys.argv)>1 else 100
seq=frames.DesktopSequence(W,H,seed=seed)
d=O.GlzDict(16*1024*1024,1)
buf=np.zeros(2*4_000_000,dtype=np.uint32)
for k in range(4):
    f=seq.next().reshape(H,-1)
    L.glzo_set_trace(buf.ctypes.data, 4_000_000)
    d.encode(O.TYPE_RGB32,W,H,W*4,[f])
    n=L.glzo_trace_count()
    t=buf[:2*n].reshape(-1,2).copy()
    pos=t[:,0].astype(np.int64); ln=(t[:,1]&0x3fffffff).astype(np.int64)
    # windows of 256 starting at a stop
    WIN=int(sys.argv[2]) if len(sys.argv)>2 else 256
    nwin=0; i=0; stops_in=[]
    nxt=pos+np.maximum(ln,1)
    while i<n:
        start=pos[i]; nwin+=1; c=0
        while i<n and pos[i]<start+WIN:
            c+=1; i+=1
        stops_in.append(c)
    st=np.array(stops_in)
    # fixed-grid blocks touched
    blk=np.unique(pos//WIN).size
    print('frame',k,'stops',n,'lit',int((ln==0).sum()),'match',int((ln>0).sum()),'windows',nwin,'stops/win mean %.1f med %d'%(st.mean(),np.median(st)),'grid blocks touched',blk, 'px in matches>=256:',int(ln[ln>=256].sum()))
