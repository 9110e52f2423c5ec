import torch, time
x = torch.empty(1512000000//2, dtype=torch.int16, device="cuda")
for name, fn in (("zero_", lambda: x.zero_()), ("fill_(1)", lambda: x.fill_(1))):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(name, "%.3f ms" % ms, "%.2f TB/s" % (x.numel() * 2 / ms / 1e9))
y = torch.empty_like(x)
for _ in range(3): y.copy_(x)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): y.copy_(x)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("copy", "%.3f ms" % ms, "%.2f TB/s (r+w)" % (2 * x.numel() * 2 / ms / 1e9))
