import torch, time
n = 1440*720*137*2
x = torch.empty(n, device='cuda'); y = torch.empty(n, device='cuda')
def t(fn, k=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(k): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/k
ms = t(lambda: x.fill_(1.0)); print("fill 1.136 GB: %.4f ms = %.2f TB/s" % (ms, n*4/ms/1e9))
ms = t(lambda: x.zero_()); print("zero_ 1.136 GB: %.4f ms = %.2f TB/s" % (ms, n*4/ms/1e9))
ms = t(lambda: torch.cuda.memset if False else x.copy_(y)); print("copy 1.136->1.136 GB: %.4f ms = %.2f TB/s total" % (ms, 2*n*4/ms/1e9))
ms = t(lambda: x.sum()); print("sum (read 1.136 GB): %.4f ms = %.2f TB/s" % (ms, n*4/ms/1e9))
