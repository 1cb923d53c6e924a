import torch, time
a=torch.randn(100000,264,device='cuda',dtype=torch.float16); b=torch.randn(100000,648,device='cuda',dtype=torch.float16)
try:
    c=torch.mm(a.t(), b, out_dtype=torch.float32); print("out_dtype ok", c.dtype, c.shape)
except Exception as e: print("out_dtype fail", repr(e)[:200])
for name,fn in [("dW f16", lambda: torch.mm(a.t(), b)), ("dH f16", lambda: torch.mm(a, torch.randn(264,640,device='cuda',dtype=torch.float16)))]:
    for _ in range(3): fn()
    torch.cuda.synchronize(); t=time.time()
    for _ in range(10): fn()
    torch.cuda.synchronize(); print(name, (time.time()-t)/10*1e3, "ms")
A=torch.randn(1270000,264,device='cuda',dtype=torch.float16); Bm=torch.randn(1270000,648,device='cuda',dtype=torch.float16); W=torch.randn(264,640,device='cuda',dtype=torch.float16)
for name,fn,fl in [("dW big", lambda: torch.mm(A.t(), Bm, out_dtype=torch.float32), 2*1.27e6*264*648), ("dH big", lambda: torch.mm(A, W), 2*1.27e6*264*640)]:
    try:
        for _ in range(2): fn()
        torch.cuda.synchronize(); t=time.time()
        for _ in range(5): fn()
        torch.cuda.synchronize(); dt=(time.time()-t)/5; print(name, dt*1e3, "ms", fl/dt/1e12, "TF/s")
    except Exception as e: print(name, "fail", repr(e)[:200])
