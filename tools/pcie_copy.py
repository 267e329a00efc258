import torch, time
x = torch.zeros(1024*1024*3, dtype=torch.float32, device="cuda")
h = torch.empty(1024*1024*3, dtype=torch.float32).pin_memory()
for _ in range(3): h.copy_(x, non_blocking=False)
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(20): h.copy_(x, non_blocking=False)
torch.cuda.synchronize()
dt=(time.perf_counter()-t0)/20
print("D2H 12.6 MB pinned: %.3f ms, %.1f GB/s" % (dt*1e3, x.numel()*4/dt/1e9))
