"""Does running the conv chain as TWO half-batch chains on two streams (store burst of one overlapping the matrix phase of
the other) beat one full-batch chain?  Forward + data-gradient launches of every v2 generator geometry, hipGraph replays."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from rave_amd import _lib as L
from rave_amd.ops import ConvGeom, _desc
sys.path.insert(0, os.path.join(ROOT, "tools"))
dev = torch.device("cuda:0")
B = 32
import importlib.util
src = open(os.path.join(ROOT, "tools", "bench_layers.py")).read()
layers = eval(src[src.index("layers = [") + len("layers = "):src.index("]\nsel")] + "]")
REP = {"unit k3 d1 C96": 4, "unit k3 d9 C96": 4, "unit k1 C96": 8, "unit k3 d3 C192": 6, "unit k1 C192": 6, "unit k3 d1 C384": 6,
       "unit k1 C384": 6, "unit k3 d3 C768": 6, "unit k1 C768": 6}


def build(nb, streams):
    """list of closures launching fwd and dgrad of every layer for batch slices of nb on the given streams"""
    calls = []
    for (name, ci, co, lin, k, st, dil, pl, pr, tr, act) in layers:
        g = ConvGeom(stride=st, dilation=dil, pad_left=pl, pad_right=pr, transposed=bool(tr), act=act, slope=0.2)
        lout = g.out_len(lin, k)
        d = _desc(g, nb, ci, co, lin, lout, k)
        r = C.byref(d)
        x = torch.randn(B, ci, lin, device=dev); y = torch.empty(B, co, lout, device=dev)
        dy = torch.randn(B, co, lout, device=dev); dx = torch.empty_like(x)
        w = torch.randn((ci, co, k) if tr else (co, ci, k), device=dev) * 0.05
        wpf = torch.empty(L.lib.rh_conv1d_packed_floats(r, 0), device=dev); wpb = torch.empty(L.lib.rh_conv1d_packed_floats(r, 1), device=dev)
        L.check(L.lib.rh_conv1d_pack_f32(r, L.ptr(w), L.ptr(wpf), L.ptr(wpb), torch.cuda.current_stream().cuda_stream))
        nf = L.lib.rh_conv1d_fwd_workspace_bytes(r); nd = L.lib.rh_conv1d_bwd_data_workspace_bytes(r)
        per = []
        for si, s in enumerate(streams):
            wsf = torch.empty(max(nf, 4) // 4, device=dev); wsd = torch.empty(max(nd, 4) // 4, device=dev)
            o = si * nb
            per.append((d, x[o:o + nb], y[o:o + nb], dy[o:o + nb], dx[o:o + nb], wpf, wpb, wsf, nf, wsd, nd, s))
        calls.append((name, per))
    return calls


def run(calls, si):
    for name, per in calls:
        d, x, y, dy, dx, wpf, wpb, wsf, nf, wsd, nd, s = per[si]
        r = C.byref(d)
        for _ in range(REP.get(name, 1)):
            L.check(L.lib.rh_conv1d_fwd_f32(r, x.data_ptr(), L.ptr(wpf), None, None, None, y.data_ptr(), L.ptr(wsf), nf, s.cuda_stream))
            L.check(L.lib.rh_conv1d_bwd_data_f32(r, dy.data_ptr(), L.ptr(wpb), x.data_ptr(), None, None, dx.data_ptr(), L.ptr(wsd), nd, s.cuda_stream))


def timed(graph, n=10):
    for _ in range(3): graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): graph.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


cap = torch.cuda.Stream()
s2 = torch.cuda.Stream()
full = build(32, [cap])
half = build(16, [cap, s2])
torch.cuda.synchronize()
with torch.cuda.stream(cap):
    run(full, 0); run(half, 0)
torch.cuda.synchronize()
g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1, stream=cap):
    run(full, 0)
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2, stream=cap):
    s2.wait_stream(cap)
    run(half, 0)
    run(half, 1)
    cap.wait_stream(s2)
g3 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g3, stream=cap):          # the two halves one after the other on ONE stream (cost of the smaller launches alone)
    half1 = [(n, [p[0], (p[1][0],) + p[1][1:11] + (cap,)]) for n, p in half]
    run(half1, 0); run(half1, 1)
print("full batch, one stream        : %.3f ms" % timed(g1))
print("two half batches, two streams : %.3f ms" % timed(g2))
print("two half batches, one stream  : %.3f ms" % timed(g3))
