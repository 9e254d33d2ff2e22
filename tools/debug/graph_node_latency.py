"""How long does a dependent kernel node of a hipGraph take when the kernel itself is empty?  (chains of N one-element
torch kernels, and of a real small HIP kernel of the library, replayed)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
dev = torch.device("cuda:0")
x = torch.zeros(1, device=dev)
big = torch.zeros(32 * 96 * 4096, device=dev)


def timed(g, n=20):
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for N in (100, 1000):
    for what, t in (("1-element add_", x), ("50 MB add_ (12.6 M elements)", big)):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3): t.add_(1.0)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(N): t.add_(1.0)
        us = timed(g)
        # eager, same stream
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(N): t.add_(1.0)
        e1.record(); torch.cuda.synchronize()
        print("%-30s chain of %4d: graph %.2f us per node, eager %.2f us per launch" % (what, N, us / N, e0.elapsed_time(e1) * 1e3 / N))
