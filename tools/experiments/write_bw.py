import torch, time
dev=torch.device("cuda:0")
for mb in (131, 268, 402, 800):
    x=torch.empty(mb*1024*1024//4, device=dev)
    y=torch.empty_like(x)
    for name,fn in (("fill",lambda: x.fill_(1.0)),("copy",lambda: y.copy_(x)),("read-sum", lambda: x.sum())):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): fn()
        e.record(); torch.cuda.synchronize()
        ms=s.elapsed_time(e)/10
        print(mb,"MB",name,"%.4f ms"%ms,"%.2f TB/s (bytes of the tensor / time)"%(mb*1.048576e6/ms/1e9))
