import torch, time
x = torch.empty(1<<33, dtype=torch.uint8, device="cuda")
y = torch.empty(1<<33, dtype=torch.uint8, device="cuda")
def t(f, n=5):
    f(); torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n
dt=t(lambda: x.fill_(3)); print("fill 8GiB: %.2f ms  %.2f TB/s write"%(dt*1e3, (1<<33)/dt/1e12))
dt=t(lambda: y.copy_(x)); print("copy 8GiB: %.2f ms  %.2f TB/s read+write"%(dt*1e3, 2*(1<<33)/dt/1e12))
xi = x.view(torch.int32)
dt=t(lambda: xi.sum()); print("sum 8GiB: %.2f ms  %.2f TB/s read"%(dt*1e3, (1<<33)/dt/1e12))
