#!/usr/bin/env python3
"""Shader clock and board power while one GEMM runs back to back (rocm-smi sampled from a side thread):
is the persistent GEMM clock/power limited at full chip load?  GPU box only."""
import sys, os, subprocess, threading, time, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rajni-vit_amd"))
import torch
from rajni_amd import ops, _native as nat

dev = "cuda"
samples = []
stop = False
def sampler():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            sclk = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", o)
            pw = re.search(r"Power \(W\): ([\d.]+)", o)
            samples.append((sclk.group(1) if sclk else "?", pw.group(1) if pw else "?"))
        except Exception as e:
            samples.append(("err", str(e)[:40]))
        time.sleep(0.3)

def run(label, fn, fl, secs=4.0):
    global samples, stop
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    samples, stop = [], False
    th = threading.Thread(target=sampler); th.start()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < secs:
        for _ in range(20):
            fn()
        torch.cuda.synchronize(); n += 20
    dt = time.perf_counter() - t0
    stop = True; th.join()
    print(f"{label:34s} {fl*n/dt/1e12:7.0f} TF  sclk/power samples: {samples[2:10]}", flush=True)

for M, N, K in [(8192, 8192, 8192), (50432, 2304, 768), (50432, 768, 3072)]:
    x = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16)
    w = (torch.rand(N, K, device=dev) * 2 - 1).to(torch.bfloat16)
    wp = ops.pack_weight(w); b = torch.zeros(N, device=dev)
    fl = 2.0 * M * N * K
    run(f"rajni {M}x{N}x{K}", lambda: ops.linear(x.view(1, M, K), wp, N, b, nat.EPI_BIAS), fl)
    run(f"hipBLASLt {M}x{N}x{K}", lambda: torch.matmul(x, w.t()), fl)
print(subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout[-1500:])
