"""whole-forward hipGraph capture: bit-identical logits and the time it buys (0.4 %: the path is not launch bound)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rajni-vit_amd"))
import torch, rajni_amd
from rajni_amd import timm_shaped as ts
sched = {3: {"keep_ratio": 0.88, "update": True}, 4: {"keep_ratio": 0.88, "update": True}, 7: {"keep_ratio": 0.80, "update": True}, 8: {"keep_ratio": 0.72, "update": True}}
cfg = ts.CONFIGS["vit_base_patch16_224"]
m = rajni_amd.RAJNIViTWrapper(ts.create_model(cfg, seed=0).to(torch.bfloat16).cuda(), sched).eval()
x = torch.randn(256, 3, 224, 224, device="cuda").to(torch.bfloat16)
for _ in range(5): y0 = m(x)
def timeit(fn, n=10, r=6):
    ts_ = []
    for _ in range(r):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); ts_.append((time.perf_counter() - t0) / n * 1e3)
    return min(ts_)
print("eager ms", timeit(lambda: m(x)))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): m(x)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    y = m(x)
g.replay(); torch.cuda.synchronize()
print("graph equal:", torch.equal(y, y0))
print("graph ms", timeit(lambda: g.replay()))
