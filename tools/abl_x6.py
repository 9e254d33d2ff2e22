"""Ablation timing of the bf16x6 conv kernel (RH_X6_ABL bits: 1 no activation loads, 2 no weight loads, 4 no MFMA,
8 no epilogue, 16 no conversion) on a few v2 layer shapes, forward only."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rave_amd import _lib as L
from rave_amd.ops import ConvGeom, _desc

dev = torch.device("cuda:0")
B = 32
layers = [
    ("unit k3 d1 C96",        96,   96, 4096, 3, 1, 1, 1, 1, 0, 1),
    ("unit k1 C96",           96,   96, 4096, 1, 1, 1, 0, 0, 0, 1),
    ("unit k3 d3 C192",      192,  192, 1024, 3, 1, 3, 3, 3, 0, 1),
    ("unit k1 C192",         192,  192, 1024, 1, 1, 1, 0, 0, 0, 1),
    ("unit k3 d1 C384",      384,  384,  256, 3, 1, 1, 1, 1, 0, 1),
    ("unit k3 d3 C768",      768,  768,   64, 3, 1, 3, 3, 3, 0, 1),
    ("down k8s4 96->192",     96,  192, 4096, 8, 4, 1, 3, 4, 0, 1),
    ("up k8s4 384->192",     384,  192,  256, 8, 4, 1, 2, 2, 1, 1),
]
abls = [int(a) for a in os.environ.get("ABLS", "0,1,2,3,4,8,16,17,7,12,31").split(",")]
s = torch.cuda.current_stream().cuda_stream
print("%-22s %7s | " % ("layer", "GFLOP") + " ".join("abl%-4d" % a for a in abls))
for (name, ci, co, lin, k, st, dil, pl, pr, tr, act) in layers:
    g = ConvGeom(stride=st, dilation=dil, pad_left=pl, pad_right=pr, transposed=bool(tr), act=act, slope=0.2)
    lout = g.out_len(lin, k)
    d = _desc(g, B, ci, co, lin, lout, k)
    r = C.byref(d)
    x = torch.randn(B, ci, lin, device=dev)
    w = torch.randn((ci, co, k) if tr else (co, ci, k), device=dev) * 0.05
    y = torch.empty(B, co, lout, device=dev)
    wpf = torch.empty(L.lib.rh_conv1d_packed_floats(r, 0), device=dev)
    wpb = torch.empty(L.lib.rh_conv1d_packed_floats(r, 1), device=dev)
    L.check(L.lib.rh_conv1d_pack_f32(r, L.ptr(w), L.ptr(wpf), L.ptr(wpb), s))
    nf = L.lib.rh_conv1d_fwd_workspace_bytes(r)
    wsf = torch.empty(max(nf, 4) // 4, device=dev)
    f = lambda: L.lib.rh_conv1d_fwd_f32(r, L.ptr(x), L.ptr(wpf), None, None, None, L.ptr(y), L.ptr(wsf), nf, s)
    flop = 2.0 * B * co * ci * k * (lin if tr else lout)
    res = []
    for a in abls:
        os.environ["RH_X6_ABL"] = str(a)
        for _ in range(2):
            L.check(f())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            L.check(f())
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 100)
    os.environ["RH_X6_ABL"] = "0"
    print("%-22s %7.2f | " % (name, flop / 1e9) + " ".join("%7.1f" % t for t in res))
