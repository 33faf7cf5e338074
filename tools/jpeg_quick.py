import sys, os, time, ctypes, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spice_amd
from spice_amd import binding as B, frames
lib=os.environ.get('JPEG_LIB')
if lib:
    gpu=B.SpiceGpu(ctypes.CDLL(lib)); gpu.init(0)
else:
    gpu=spice_amd.load(0)
W,H=3840,2160
F=16
seq=frames.DesktopSequence(W,H,seed=3) if os.environ.get('CONTENT')=='desktop' else frames.VideoSequence(W,H,seed=7)
fr=[seq.next() for _ in range(4)]
fmt,bpp=B.BITMAP_FMT_32BIT,4
if os.environ.get('FMT')=='24': fr=[frames.to_bgr24(f) for f in fr]; fmt,bpp=B.BITMAP_FMT_24BIT,3
if os.environ.get('FMT')=='16': fr=[frames.to_rgb16(f) for f in fr]; fmt,bpp=B.BITMAP_FMT_16BIT,2
dev=[gpu.to_device(f) for f in fr]
cap=W*H
outs=[gpu.malloc(cap) for _ in range(F)]
dct=int(sys.argv[1]) if len(sys.argv)>1 else B.DCT_IFAST
q=70 if dct==B.DCT_IFAST else 85
items=[dict(pixels=dev[f%4],format=fmt,width=W,height=H,stride=W*bpp,quality=q,dct=dct,out=outs[f],out_capacity=cap) for f in range(F)]
try:
    B.jpeg_encode_batch(gpu,items)
except Exception as e:
    print('first batch', e)
gpu.L.spice_gpu_timing_enable(0 if os.environ.get('NOTIMING') else 1); gpu.L.spice_gpu_timing_reset()
t0=time.time()
for _ in range(20):
    try: B.jpeg_encode_batch(gpu,items)
    except Exception as e: pass
dt=(time.time()-t0)/20
print(lib, 'ms/batch %.3f'%(dt*1e3), {k:round(v[1]/max(1,v[0]),3) for k,v in gpu.kernel_times().items()})
