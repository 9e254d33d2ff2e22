"""Throughput of the GPU data feed (rave_amd/data.py): batches of 32 x 65536 from 64 resident items."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rave_amd import data as D
dev = torch.device("cuda:0")
pcm = torch.randint(-20000, 20000, (64, 1, 4 * 65536), dtype=torch.int16, device=dev)
feed = D.GpuBatchFeed(pcm, seed=0)
for _ in range(3):
    feed.sample(32, 65536)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n = 10
for _ in range(n):
    feed.sample(32, 65536)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
print(f"feed: {ms:.3f} ms per batch of 32 x 65536 = {32 * 65536 / ms / 1e3:.1f} M samples/s (noise draw included)")
