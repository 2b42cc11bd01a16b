"""host-side cost of one RAJNIViTWrapper.forward call (plan lookup, weight-cache check, the native call), GPU box only."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rajni-vit_amd"))
import torch, rajni_amd
from rajni_amd import timm_shaped as ts
sched = {3: {"keep_ratio": 0.88, "update": True}, 4: {"keep_ratio": 0.88, "update": True}, 7: {"keep_ratio": 0.80, "update": True}, 8: {"keep_ratio": 0.72, "update": True}}
cfg = ts.CONFIGS["vit_base_patch16_224"]
m = rajni_amd.RAJNIViTWrapper(ts.create_model(cfg, seed=0).to(torch.bfloat16).cuda(), sched).eval()
x = torch.randn(256, 3, 224, 224, device="cuda").to(torch.bfloat16)
for _ in range(5): m(x)
torch.cuda.synchronize()
# host time of one forward call (async launches)
ts_ = []
for _ in range(20):
    torch.cuda.synchronize(); t0 = time.perf_counter(); m(x); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ts_.append((t1 - t0, t2 - t0))
print("host call time us: min %.0f med %.0f | sync-to-sync ms: min %.3f" % (min(a for a, _ in ts_) * 1e6, sorted(a for a, _ in ts_)[10] * 1e6, min(b for _, b in ts_) * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(50): m(x)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
