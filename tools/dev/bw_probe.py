import torch, time
x = torch.empty(1<<28, dtype=torch.float32, device="cuda")   # 1 GiB
y = torch.empty_like(x)
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); s=torch.cuda.Event(True); e=torch.cuda.Event(True); s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n
ms=t(lambda: x.fill_(1.0)); print("fill  1 GiB: %.3f ms  %.2f TB/s write" % (ms, 1.0737/ms))
ms=t(lambda: x.zero_()); print("zero  1 GiB: %.3f ms  %.2f TB/s write" % (ms, 1.0737/ms))
ms=t(lambda: y.copy_(x)); print("copy  1 GiB: %.3f ms  %.2f TB/s r+w" % (ms, 2*1.0737/ms))
ms=t(lambda: x.sum()); print("sum   1 GiB: %.3f ms  %.2f TB/s read" % (ms, 1.0737/ms))
ms=t(lambda: torch.add(x, 1.0, out=y)); print("add   1 GiB: %.3f ms  %.2f TB/s r+w" % (ms, 2*1.0737/ms))
