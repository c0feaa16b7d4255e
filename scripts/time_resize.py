#!/usr/bin/env python3
"""Device time of the batched Lanczos stretch (HIP events) + a digest of its output: A/B with DD_LANCZOS_FUSED=0."""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deepdish_amd._lib import lib, check
from deepdish_amd.runtime import default_context, ptr

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 384
H, W, h, w = (int(v) for v in sys.argv[2:6]) if len(sys.argv) > 5 else (480, 640, 300, 300)
ctx = default_context()
g = torch.Generator(device='cuda'); g.manual_seed(0)
src = torch.randint(0, 256, (batch, H, W, 3), dtype=torch.uint8, device='cuda', generator=g)
dst = torch.empty((batch, h, w, 3), dtype=torch.uint8, device='cuda')
def run():
    check(lib().dd_resize_lanczos_batch(ctx.handle, ptr(src), batch, H, W, 3, 1, ptr(dst), h, w, None))
for _ in range(3): run()
ctx.sync(); torch.cuda.synchronize()
ts = []
for _ in range(20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); ctx.sync()
    import time; t0 = time.perf_counter()
    run(); ctx.sync()
    ts.append((time.perf_counter() - t0) * 1e6)
sha = hashlib.sha256(dst.cpu().numpy().tobytes()).hexdigest()[:16]
print(f'lanczos {batch} x {H}x{W} -> {h}x{w}: min {min(ts):.1f} us  median {sorted(ts)[len(ts)//2]:.1f} us (host clock around launch + sync)  sha {sha}')
