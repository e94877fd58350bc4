import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pointcloudmatters_amd import _lib
L=_lib.load(); dev="cuda"; f32=dict(dtype=torch.float32, device=dev)
for n_par in (24_100_000//64*64, 255_600_000//64*64):
    p,g=torch.randn(n_par,**f32),torch.randn(n_par,**f32)*1e-3; m,v=torch.zeros(n_par,**f32),torch.zeros(n_par,**f32)
    pb=torch.empty(n_par,dtype=torch.bfloat16,device=dev)
    hyper=torch.tensor([5e-5,0.9,0.999,1e-8,0.05,0.1,0.0316,0.5,1.0,0.0],**f32); parts=torch.zeros(L.pcm_optim_partials_capacity(),**f32)
    npart=ctypes.c_int(0); st=torch.cuda.current_stream().cuda_stream
    L.pcm_grad_sumsq_hip(n_par,g.data_ptr(),parts.data_ptr(),ctypes.addressof(npart),st)
    def run(): assert L.pcm_adamw_flat_hip(n_par,p.data_ptr(),g.data_ptr(),m.data_ptr(),v.data_ptr(),hyper.data_ptr(),parts.data_ptr(),npart.value,0,pb.data_ptr(),st)==0
    for _ in range(3): run()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    ts=[]
    for _ in range(20):
        e0.record(); run(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); t=ts[len(ts)//2]
    print(f"n={n_par/1e6:.1f}M adamw {t*1e3:.1f} us  {30*n_par/t/1e6:.0f} GB/s  frac {30*n_par/t/1e6/8000:.3f}")
    def run2(): assert L.pcm_grad_sumsq_hip(n_par,g.data_ptr(),parts.data_ptr(),ctypes.addressof(npart),st)==0
    for _ in range(3): run2()
    ts=[]
    for _ in range(20):
        e0.record(); run2(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); t=ts[len(ts)//2]
    print(f"n={n_par/1e6:.1f}M sumsq {t*1e3:.1f} us  {4*n_par/t/1e6:.0f} GB/s  frac {4*n_par/t/1e6/8000:.3f}")
