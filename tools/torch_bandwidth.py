import torch, time
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n*1e-3
for mb in (64, 256, 512, 1024, 2048):
    n = mb*1024*1024//2
    x = torch.empty(n, dtype=torch.bfloat16, device='cuda').normal_(); y = torch.empty_like(x); z = torch.empty_like(x)
    tc = t(lambda: y.copy_(x)); ta = t(lambda: torch.add(x, y, out=z)); ts = t(lambda: x.sum())
    print(f"{mb} MB: copy {2*mb/1024/tc/1e3*1.0737:.2f} TB/s  add(2r+1w) {3*mb/1024/ta/1e3*1.0737:.2f} TB/s  sum(read) {mb/1024/ts/1e3*1.0737:.2f} TB/s")
