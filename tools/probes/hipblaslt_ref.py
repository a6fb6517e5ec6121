import torch, time
def t(M,N,K):
    a=torch.randn(M,K,device='cuda',dtype=torch.bfloat16); b=torch.randn(N,K,device='cuda',dtype=torch.bfloat16)
    for _ in range(5): (a@b.t())
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): c=a@b.t()
    e.record(); torch.cuda.synchronize()
    ms=s.elapsed_time(e)/20
    print(M,N,K,'%.3f ms %.0f TF'%(ms, 2*M*N*K/ms/1e9))
t(8*256*256,256,2304)
t(8*128*128,512,4608)
t(8*64*64,1024,9216//1)
t(8*32*32,2048,4608*4)
t(8*256*256,256,256)
t(8*256*256,64,256)
t(8*256*256,256,64)
t(8192,8192,8192)
