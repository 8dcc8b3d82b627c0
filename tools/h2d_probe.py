import torch, time
h = torch.empty(796<<20, dtype=torch.uint8).pin_memory()
d = torch.empty_like(h, device="cuda")
for _ in range(2): d.copy_(h, non_blocking=True)
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(5): d.copy_(h, non_blocking=True)
torch.cuda.synchronize(); el=(time.perf_counter()-t)/5
print("H2D GB/s", h.numel()/el/1e9, "ms", el*1e3)
