"""GPU micro-benchmark: time every distinct conv geometry of the v2 generator path (B=32, T=65536)
through the C ABI: forward, data-gradient, weight-gradient.  Prints TFLOP/s per layer."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rave_amd import _lib as L
from rave_amd.ops import ConvGeom, _desc

dev = torch.device("cuda:0")
B = int(os.environ.get("B", 32))
# (name, cin, cout, L_in, k, stride, dil, pad_l, pad_r, transposed, act)
layers = [
    ("stem k7 16->96",        16,   96, 4096, 7, 1, 1, 3, 3, 0, 0),
    ("unit k3 d1 C96",        96,   96, 4096, 3, 1, 1, 1, 1, 0, 1),
    ("unit k3 d9 C96",        96,   96, 4096, 3, 1, 9, 9, 9, 0, 1),
    ("unit k1 C96",           96,   96, 4096, 1, 1, 1, 0, 0, 0, 1),
    ("down k8s4 96->192",     96,  192, 4096, 8, 4, 1, 3, 4, 0, 1),
    ("unit k3 d3 C192",      192,  192, 1024, 3, 1, 3, 3, 3, 0, 1),
    ("unit k1 C192",         192,  192, 1024, 1, 1, 1, 0, 0, 0, 1),
    ("down k8s4 192->384",   192,  384, 1024, 8, 4, 1, 3, 4, 0, 1),
    ("unit k3 d1 C384",      384,  384,  256, 3, 1, 1, 1, 1, 0, 1),
    ("unit k1 C384",         384,  384,  256, 1, 1, 1, 0, 0, 0, 1),
    ("down k8s4 384->768",   384,  768,  256, 8, 4, 1, 3, 4, 0, 1),
    ("unit k3 d3 C768",      768,  768,   64, 3, 1, 3, 3, 3, 0, 1),
    ("unit k1 C768",         768,  768,   64, 1, 1, 1, 0, 0, 0, 1),
    ("down k4s2 768->1536",  768, 1536,   64, 4, 2, 1, 1, 2, 0, 1),
    ("head k3 1536->256",   1536,  256,   32, 3, 1, 1, 1, 1, 0, 1),
    ("dec in k3 128->1536",  128, 1536,   32, 3, 1, 1, 1, 1, 0, 0),
    ("up k4s2 1536->768",   1536,  768,   32, 4, 2, 1, 1, 1, 1, 1),
    ("up k8s4 768->384",     768,  384,   64, 8, 4, 1, 2, 2, 1, 1),
    ("up k8s4 384->192",     384,  192,  256, 8, 4, 1, 2, 2, 1, 1),
    ("up k8s4 192->96",      192,   96, 1024, 8, 4, 1, 2, 2, 1, 1),
    ("out k7 96->32",         96,   32, 4096, 7, 1, 1, 3, 3, 0, 1),
]
sel = os.environ.get("ONLY")
s = torch.cuda.current_stream().cuda_stream
print("%-24s %9s | %8s %7s | %8s %7s | %8s %7s" % ("layer", "GFLOP", "fwd us", "TF/s", "dgrad us", "TF/s", "wgrad us", "TF/s"))
tot = [0.0, 0.0, 0.0, 0.0]
for (name, ci, co, lin, k, st, dil, pl, pr, tr, act) in layers:
    if sel and sel not in name:
        continue
    g = ConvGeom(stride=st, dilation=dil, pad_left=pl, pad_right=pr, transposed=bool(tr), act=act, slope=0.2)
    lout = g.out_len(lin, k)
    d = _desc(g, B, ci, co, lin, lout, k)
    r = C.byref(d)
    x = torch.randn(B, ci, lin, device=dev)
    w = torch.randn((ci, co, k) if tr else (co, ci, k), device=dev) * 0.05
    y = torch.empty(B, co, lout, device=dev)
    dy = torch.randn(B, co, lout, device=dev)
    dx = torch.empty_like(x)
    dw = torch.empty_like(w)
    wpf = torch.empty(L.lib.rh_conv1d_packed_floats(r, 0), device=dev)
    wpb = torch.empty(L.lib.rh_conv1d_packed_floats(r, 1), device=dev)
    L.check(L.lib.rh_conv1d_pack_f32(r, L.ptr(w), L.ptr(wpf), L.ptr(wpb), s))
    nws = L.lib.rh_conv1d_workspace_bytes(r)
    ws = torch.empty(max(nws, 4) // 4, device=dev)
    nf = L.lib.rh_conv1d_fwd_workspace_bytes(r); nd = L.lib.rh_conv1d_bwd_data_workspace_bytes(r)
    wsf = torch.empty(max(nf, 4) // 4, device=dev); wsd = torch.empty(max(nd, 4) // 4, device=dev)
    # range slots (f16 build, include/rave_hip.h: rh_x6_set_ranges): the inputs' are filled once, the outputs' are written by
    # every launch as in the training step (never re-zeroed here: atomicMax of the same values)
    use_r = L.lib.rh_x6_uses_ranges() == 1
    sl = torch.zeros(4, L.lib.rh_x6_range_words(), device=dev, dtype=torch.int32)
    if use_r:
        L.check(L.lib.rh_amax_f32(L.ptr(x), x.numel(), L.ptr(sl[0]), s))
        L.check(L.lib.rh_amax_f32(L.ptr(dy), dy.numel(), L.ptr(sl[1]), s))
    arm = (lambda a, b, o: L.lib.rh_x6_set_ranges(L.ptr(a), L.ptr(b), L.ptr(o), None) and 0) if use_r else (lambda a, b, o: 0)
    fns = [
        lambda: arm(None, sl[0], sl[2]) or L.lib.rh_conv1d_fwd_f32(r, L.ptr(x), L.ptr(wpf), None, None, None, L.ptr(y), L.ptr(wsf), nf, s),
        lambda: arm(None, sl[1], sl[3]) or L.lib.rh_conv1d_bwd_data_f32(r, L.ptr(dy), L.ptr(wpb), L.ptr(x), None, None, L.ptr(dx), L.ptr(wsd), nd, s),
        lambda: arm(sl[1], sl[0], None) or L.lib.rh_conv1d_bwd_weight_f32(r, L.ptr(dy), L.ptr(x), None, L.ptr(dw), None, L.ptr(ws), nws, s),
    ]
    flop = 2.0 * B * co * ci * k * (lin if tr else lout)
    res = []
    for f in fns:
        for _ in range(2):
            L.check(f())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 5
        e0.record()
        for _ in range(n):
            L.check(f())
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        res.append(us)
    if os.environ.get("CONCURRENT"):
        # dgrad on the current stream and wgrad on a side stream at the same time (both only need dy and x)
        side = torch.cuda.Stream()
        s2 = side.cuda_stream
        fw2 = lambda: arm(sl[1], sl[0], None) or L.lib.rh_conv1d_bwd_weight_f32(r, L.ptr(dy), L.ptr(x), None, L.ptr(dw), None, L.ptr(ws), nws, s2)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 5
        e0.record()
        for _ in range(n):
            side.wait_stream(torch.cuda.current_stream())
            L.check(fns[1]()); L.check(fw2())
            torch.cuda.current_stream().wait_stream(side)
        e1.record()
        torch.cuda.synchronize()
        both = e0.elapsed_time(e1) * 1e3 / n
        print("    dgrad || wgrad on two streams: %.1f us (serial %.1f)" % (both, res[1] + res[2]))
    tot[0] += flop; tot[1] += res[0]; tot[2] += res[1]; tot[3] += res[2]
    print("%-24s %9.2f | %8.1f %7.1f | %8.1f %7.1f | %8.1f %7.1f" % (name, flop / 1e9, res[0], flop / res[0] / 1e6, res[1], flop / res[1] / 1e6, res[2], flop / res[2] / 1e6))
print("TOTAL (one of each)      %9.2f | %8.1f %7.1f | %8.1f %7.1f | %8.1f %7.1f" % (tot[0] / 1e9, tot[1], tot[0] / tot[1] / 1e6, tot[2], tot[0] / tot[2] / 1e6, tot[3], tot[0] / tot[3] / 1e6))
