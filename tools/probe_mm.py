import torch, time
def bench(name, fn, fl):
    try:
        for _ in range(3): fn()
        torch.cuda.synchronize(); t=time.time()
        for _ in range(20): fn()
        torch.cuda.synchronize(); dt=(time.time()-t)/20; print(f"{name}: {dt*1e6:.1f} us  {fl/dt/1e12:.1f} TF/s")
    except Exception as e: print(name, "fail", repr(e)[:200])
M=12032
for N,K in [(1024,256),(256,1024),(256,256),(768,256)]:
    dY=torch.randn(M,N,device='cuda',dtype=torch.bfloat16); X=torch.randn(M,K,device='cuda',dtype=torch.bfloat16)
    fl=2*M*N*K
    print(f"--- dW[{N},{K}] = dY^T X, M={M}")
    bench("TN mm              ", lambda: torch.mm(dY.t(), X), fl)
    bench("TN mm f32 out      ", lambda: torch.mm(dY.t(), X, out_dtype=torch.float32), fl)
    def tr_bmm(S):
        dYt=dY.t().contiguous().view(N,S,M//S).transpose(0,1); Xt=X.t().contiguous().view(K,S,M//S).transpose(0,1)
        return torch.bmm(dYt, Xt.transpose(1,2), out_dtype=torch.float32).sum(0)
    for S in (4,8,16):
        bench(f"transposes+bmm S={S:2d} ", lambda: tr_bmm(S), fl)
    def view_bmm(S):
        a=dY.view(S,M//S,N).transpose(1,2); b=X.view(S,M//S,K)
        return torch.bmm(a,b,out_dtype=torch.float32).sum(0)
    for S in (4,8,16,47):
        bench(f"view bmm (TN) S={S:2d}   ", lambda: view_bmm(S), fl)
    bench("transpose only      ", lambda: (dY.t().contiguous(), X.t().contiguous()), 0)
