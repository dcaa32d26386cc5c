"""E engines x B pairs on E HIP streams, ticks launched round-robin: does overlapping the
latency-bound kernels of one engine with the MFMA kernels of another raise throughput?
    python tools/tick_bench2.py E B [ticks] [G]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deformationpyramid_amd.config import load_config
from deformationpyramid_amd.engine import BatchedEngine
from deformationpyramid_amd.registration import Registration
from deformationpyramid_amd.synthetic import synthetic_pair

E = int(sys.argv[1]); B = int(sys.argv[2]); ticks = int(sys.argv[3]) if len(sys.argv) > 3 else 32
G = int(sys.argv[4]) if len(sys.argv) > 4 else None
torch.set_num_threads(8)
dev = torch.device("cuda:0")
cfg = load_config(os.path.join(ROOT, "config", "NDP.yaml"), device=0)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _modes import from_env
model = Registration(cfg, **from_env())
engs, streams = [], []
for e in range(E):
    preps = []
    for i in range(B):
        s, t, _, _ = synthetic_pair(e * B + i)
        preps.append(model._prepare(s.to(dev), t.to(dev), None))
    eng = BatchedEngine(preps[0].desc, model._opt_config(False), B, 2048, 2048, dev, G=G)
    for b, p in enumerate(preps):
        eng.load_jobs([p.load_job(b)])
    engs.append(eng); streams.append(torch.cuda.Stream(dev))
torch.cuda.synchronize()
def run(n, chunk=1):
    for k in range(0, n, chunk):
        for eng, st in zip(engs, streams):
            with torch.cuda.stream(st):
                eng.run_ticks(chunk)
run(4); torch.cuda.synchronize()
for chunk in (1, 4):
    t0 = time.perf_counter(); run(ticks, chunk); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"E={E} B={B} G={engs[0].G} chunk={chunk}: {1e3*dt/ticks:.4f} ms per round of ticks, {1e6*dt/ticks/(E*B):.2f} us per pair-iteration")
