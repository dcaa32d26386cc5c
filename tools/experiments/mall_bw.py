"""Does a working set that fits the 256 MB Infinity Cache stream faster than one that does not?  fill (write), sum (read) and a
write-then-read round trip of N megabytes, repeated: effective TB/s by size."""
import torch, time
dev = torch.device("cuda:0")
def bench(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
for mb in (32, 64, 128, 192, 256, 384, 512, 1024, 2048):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
    a.fill_(1.0)
    tw = bench(lambda: a.fill_(2.0))
    tr = bench(lambda: a.sum())
    tc = bench(lambda: b.copy_(a))
    def rt():
        a.fill_(3.0); a.sum()
    trt = bench(rt)
    print(f"{mb:5d} MB: fill {mb/1024/1024/tw*1.048576:.2f} TB/s  sum {mb/1024/1024/tr*1.048576:.2f} TB/s  copy (r+w) {2*mb/1024/1024/tc*1.048576:.2f} TB/s  fill+sum {2*mb/1024/1024/trt*1.048576:.2f} TB/s", flush=True)
