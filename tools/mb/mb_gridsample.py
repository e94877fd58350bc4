import sys, os, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pointcloudmatters_amd.bc.gpu_transforms import GpuPcdPipeline, voxel_keys
dev="cuda"
for b, n in ((8, 16384), (16, 16384), (64, 16384)):
    rng=np.random.default_rng(0)
    coord=np.empty((b*n,3),np.float32); coord[:,:2]=rng.uniform(-0.4,0.4,(b*n,2)); coord[:,2]=rng.uniform(0.005,0.4,b*n)
    coord=torch.from_numpy(coord).to(dev); color=torch.randint(0,256,(b*n,3),device=dev).float()
    off=torch.arange(1,b+1,device=dev)*n
    pipe=GpuPcdPipeline(0.005)
    for _ in range(3): out=pipe(coord,color,off)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(10): out=pipe(coord,color,off)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/10*1e3
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): voxel_keys(coord,off,0.005)
    e1.record(); torch.cuda.synchronize()
    print(f"b={b} n={n}: pipeline {dt:.2f} ms/batch ({b*n/dt/1e3:.1f} M points/s), voxel keys {e0.elapsed_time(e1)/20*1e3:.0f} us, kept {out['coord'].shape[0]} of {b*n}")
