"""Eager render_rays timing in bf16 and bf16x3 at 1024 rays x 64 samples (no graph, no bank): a smoke-level number."""
import time, torch, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from satnerf_amd import data as O  # synthetic rays / default args (the oracle is test infrastructure)
from satnerf_amd import rendering, ops
from satnerf_amd.models import load_model
dev='cuda:0'
args=O.default_args()
m=load_model(args).to(dev)
emb=torch.nn.Embedding(30,4).to(dev)
rays,ts=O.synthetic_rays(1024); rays=rays.to(dev); ts=ts.to(dev)
for mode in ('bf16','bf16x3'):
    args.mlp_mode=mode
    hi,lo,l0=m.packed(mode)
    z=ops.ray_sample(rays, torch.rand(1024,64,device=dev),64)
    for _ in range(5): ops.satnerf_mlp(rays[:,0:3],rays[:,3:6],rays[:,8:11],z,emb.weight.data,ts,65536,64,256,4,mode,hi,lo,l0)
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ops.satnerf_mlp(rays[:,0:3],rays[:,3:6],rays[:,8:11],z,emb.weight.data,ts,65536,64,256,4,mode,hi,lo,l0)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/50
    print(f"{mode}: mlp kernel {ms*1e3:.1f} us -> {1024/ms*1e3/1e6:.2f} M rays/s, {86.44e9/ms/1e9:.1f} TFLOP/s")
    with torch.no_grad():
        for _ in range(5): rendering.render_rays({'coarse':m,'t':emb},args,rays,ts)
        torch.cuda.synchronize(); t0=time.time()
        for _ in range(50): rendering.render_rays({'coarse':m,'t':emb},args,rays,ts)
        torch.cuda.synchronize(); dt=(time.time()-t0)/50
    print(f"{mode}: render_rays {dt*1e6:.1f} us -> {1024/dt/1e6:.2f} M rays/s")
