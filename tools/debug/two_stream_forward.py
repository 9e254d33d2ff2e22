"""Would interleaving INDEPENDENT launches hide the per-launch fixed cost of the forward chain?  The no-grad v2 forward at
batch 32 as one chain, against the same 32 clips as two independent batch-16 chains on two streams (recorded into ONE hipGraph
with a fork / join, replayed), against batch 16 alone.  Timing only: the two chains share the library's split-K scratch cache, so
the concurrent run's VALUES are not checked.   python tools/debug/two_stream_forward.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from rave_amd import model as M, ops as R

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = M.build_v2().to(dev).train()
x32 = (0.3 * torch.randn(32, 1, 65536)).clamp(-1, 1).to(dev)
xa, xb = x32[:16].contiguous(), x32[16:].contiguous()


def fwd(x):
    return m.decode(m.encoder.reparametrize(m.encode(x))[0])


def timed_graph(record, n=20):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        record()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with torch.no_grad():
    m.prepare_weights(reuse=True)
    m.set_phase_flags_eagerly()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):            # eager warm-up of every shape, on a side stream
        for x in (x32, xa, xb):
            fwd(x)
    torch.cuda.synchronize()

    def one32():
        R.range_reset(dev)
        fwd(x32)

    def one16():
        R.range_reset(dev)
        fwd(xa)

    def two16_serial():
        R.range_reset(dev)
        fwd(xa)
        fwd(xb)

    s2 = torch.cuda.Stream()

    def two16_concurrent():
        R.range_reset(dev)
        cur = torch.cuda.current_stream()
        s2.wait_stream(cur)
        with torch.cuda.stream(s2):
            fwd(xb)
        fwd(xa)
        cur.wait_stream(s2)

    t32 = timed_graph(one32)
    t16 = timed_graph(one16)
    t2s = timed_graph(two16_serial)
    t2c = timed_graph(two16_concurrent)
    print(f"forward, hipGraph replay: batch 32 one chain {t32:.3f} ms | batch 16 one chain {t16:.3f} ms | 2 x batch 16 in series {t2s:.3f} ms | "
          f"2 x batch 16 on two streams {t2c:.3f} ms  ({t2c / t32:.2f} x the batch-32 chain)")
