import torch
x=torch.empty(287*1024*1024//4,device='cuda')
y=torch.empty_like(x)
def t(fn,n=20):
    for _ in range(3): fn()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
b=x.numel()*4
us=t(lambda: x.fill_(1.0)); print('fill  %.1f us  %.2f TB/s write'%(us,b/us/1e6))
us=t(lambda: x.zero_()); print('zero  %.1f us  %.2f TB/s write'%(us,b/us/1e6))
us=t(lambda: y.copy_(x)); print('copy  %.1f us  %.2f TB/s read+write'%(us,2*b/us/1e6))
us=t(lambda: x.sum()); print('sum   %.1f us  %.2f TB/s read'%(us,b/us/1e6))
