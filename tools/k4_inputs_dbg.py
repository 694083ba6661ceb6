import sys, numpy as np, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import kpeg_testlib as T, libkpeg_amd as K
for (w,h,q,sig,mode,name) in ((1920,1080,95,6.0,1,'noise q95'),(1920,1080,75,6.0,0,'field q75'),(3840,2160,90,20.0,0,'field q90 s20')):
    data = T.synth_jpeg(w,h,seed=3,quality=q,sigma=sig,mode=mode)
    p = T.oracle_parse(data); f = T.make_frame(p)
    ctx = K.Context(0); ctx.set_profiling(True)
    d_scan = torch.frombuffer(bytearray(p.scan), dtype=torch.uint8).cuda()
    d_rgb = torch.empty((h,w,3),dtype=torch.uint8,device='cuda')
    for i in range(6):
        ctx.decode_scan_dev(f, d_scan.data_ptr(), len(p.scan), d_rgb.data_ptr()); ctx.sync()
    t = ctx.timings()
    print(name, len(p.scan), {k: round(v,4) if isinstance(v,float) else v for k,v in t.items()})
    ctx.close()
