import torch, time
n = 8_600_000
h = torch.empty(n, dtype=torch.uint8, pin_memory=True); d = torch.empty(n, dtype=torch.uint8, device="cuda")
for _ in range(5): d.copy_(h, non_blocking=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): d.copy_(h, non_blocking=True)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print("H2D 8.6 MB pinned: %.3f ms = %.1f GB/s" % (ms, n / ms / 1e6))
small = torch.empty(110_000, dtype=torch.uint8, pin_memory=True); ds = torch.empty(110_000, dtype=torch.uint8, device="cuda")
e0.record()
for _ in range(20): small.copy_(ds, non_blocking=True)
e1.record(); torch.cuda.synchronize()
print("D2H 110 KB: %.3f ms" % (e0.elapsed_time(e1) / 20))
