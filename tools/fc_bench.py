import sys, os, torch
sys.path.insert(0, os.getcwd())
from slam_amd.network import encoder_ops as ops
for (M,N,K) in ((128,4096,25088),(128,4096,4096),(128,25088,4096)):
    a=torch.randn(M,K,device="cuda").to(torch.bfloat16); b=torch.randn(N,K,device="cuda").to(torch.bfloat16)
    out=torch.empty(M,N,device="cuda")
    for _ in range(3): ops.gemm_nt(a,b,out_f32=out)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.gemm_nt(a,b,out_f32=out)
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)/20*1e3
    print(M,N,K,"%.1f us %.0f TFLOP/s %.0f GB/s weights"%(us, 2.0*M*N*K/us/1e6, 2.0*N*K/us/1e3))
