import torch, time
def bench(name, fn, fl):
    try:
        for _ in range(2): fn()
        torch.cuda.synchronize(); t=time.time()
        for _ in range(5): fn()
        torch.cuda.synchronize(); dt=(time.time()-t)/5; print(f"{name}: {dt*1e3:.3f} ms  {fl/dt/1e12:.1f} TF/s")
    except Exception as e: print(name, "fail", repr(e)[:200])
K=1275392
GT=torch.randn(264,K,device='cuda',dtype=torch.float16); HT=torch.randn(648,K,device='cuda',dtype=torch.float16)
G=GT.t().contiguous(); Hd=HT.t().contiguous()
fl=2*264*648*K
bench("TN  G.t() @ hid          ", lambda: torch.mm(G.t(), Hd, out_dtype=torch.float32), fl)
bench("NT  GT @ HT.t()          ", lambda: torch.mm(GT, HT.t(), out_dtype=torch.float32), fl)
bench("NT f16 out               ", lambda: torch.mm(GT, HT.t()), fl)
GTb=GT.bfloat16(); HTb=HT.bfloat16()
bench("NT bf16                  ", lambda: torch.mm(GTb, HTb.t(), out_dtype=torch.float32), fl)
# split-K by hand: bmm over 64 chunks
S=64; Kc=K//S
GTc=GT[:,:S*Kc].reshape(264,S,Kc).permute(1,0,2).contiguous(); HTc=HT[:,:S*Kc].reshape(648,S,Kc).permute(1,0,2).contiguous()
bench("bmm split-K 64           ", lambda: torch.bmm(GTc, HTc.transpose(1,2), out_dtype=torch.float32).sum(0), fl)
