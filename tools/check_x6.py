"""GPU check + micro-benchmark of the bf16x6 convolution kernels (conv_x6_kernel.inc) against the exact-f32 MFMA
kernels (RH_CONV_X6=0) through the C ABI: forward and data gradient of every conv geometry of the v2 generator path
(B=32, T=65536) plus edge cases (ragged lengths, batch folding, bias / residual / activation-derivative epilogues,
MPD-style inner > 1).  Prints max relative L2 difference and TFLOP/s of both paths."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rave_amd import _lib as L
from rave_amd.ops import ConvGeom, _desc

dev = torch.device("cuda:0")
B = int(os.environ.get("B", 32))
NIT = int(os.environ.get("NIT", 5))
# (name, cin, cout, L_in, k, stride, dil, pad_l, pad_r, transposed, act, batch, inner)
layers = [
    ("stem k7 16->96",        16,   96, 4096, 7, 1, 1, 3, 3, 0, 0, B, 1),
    ("unit k3 d1 C96",        96,   96, 4096, 3, 1, 1, 1, 1, 0, 1, B, 1),
    ("unit k3 d9 C96",        96,   96, 4096, 3, 1, 9, 9, 9, 0, 1, B, 1),
    ("unit k1 C96",           96,   96, 4096, 1, 1, 1, 0, 0, 0, 1, B, 1),
    ("down k8s4 96->192",     96,  192, 4096, 8, 4, 1, 3, 4, 0, 1, B, 1),
    ("unit k3 d3 C192",      192,  192, 1024, 3, 1, 3, 3, 3, 0, 1, B, 1),
    ("unit k1 C192",         192,  192, 1024, 1, 1, 1, 0, 0, 0, 1, B, 1),
    ("down k8s4 192->384",   192,  384, 1024, 8, 4, 1, 3, 4, 0, 1, B, 1),
    ("unit k3 d1 C384",      384,  384,  256, 3, 1, 1, 1, 1, 0, 1, B, 1),
    ("unit k1 C384",         384,  384,  256, 1, 1, 1, 0, 0, 0, 1, B, 1),
    ("down k8s4 384->768",   384,  768,  256, 8, 4, 1, 3, 4, 0, 1, B, 1),
    ("unit k3 d3 C768",      768,  768,   64, 3, 1, 3, 3, 3, 0, 1, B, 1),
    ("unit k1 C768",         768,  768,   64, 1, 1, 1, 0, 0, 0, 1, B, 1),
    ("down k4s2 768->1536",  768, 1536,   64, 4, 2, 1, 1, 2, 0, 1, B, 1),
    ("head k3 1536->256",   1536,  256,   32, 3, 1, 1, 1, 1, 0, 1, B, 1),
    ("dec in k3 128->1536",  128, 1536,   32, 3, 1, 1, 1, 1, 0, 0, B, 1),
    ("up k4s2 1536->768",   1536,  768,   32, 4, 2, 1, 1, 1, 1, 1, B, 1),
    ("up k8s4 768->384",     768,  384,   64, 8, 4, 1, 2, 2, 1, 1, B, 1),
    ("up k8s4 384->192",     384,  192,  256, 8, 4, 1, 2, 2, 1, 1, B, 1),
    ("up k8s4 192->96",      192,   96, 1024, 8, 4, 1, 2, 2, 1, 1, B, 1),
    ("out k7 96->32",         96,   32, 4096, 7, 1, 1, 3, 3, 0, 1, B, 1),
    # edge cases
    ("ragged k3 d2 C48->80 L1000 B3",   48,  80, 1000, 3, 1, 2, 2, 2, 0, 1, 3, 1),
    ("ragged causal k3 d3 C32 L333 B5", 32,  32,  333, 3, 1, 3, 6, 0, 0, 1, 5, 1),
    ("ragged k7 C16->40 L77 B2",        16,  40,   77, 7, 1, 1, 3, 3, 0, 0, 2, 1),
    ("ragged down k4s2 C24->56 L250 B3", 24, 56,  250, 4, 2, 1, 1, 2, 0, 1, 3, 1),
    ("ragged down k8s4 C20->96 L1001 B2", 20, 96, 1001, 8, 4, 1, 3, 4, 0, 1, 2, 1),
    ("msd k15s4 C32->128 L4099 B2",     32, 128, 4099, 15, 4, 1, 7, 7, 0, 0, 2, 1),
    ("ragged up k8s4 C64->24 L67 B3",   64,  24,   67, 8, 4, 1, 2, 2, 1, 1, 3, 1),
    ("ragged up k4s2 C48->36 L129 B2",  48,  36,  129, 4, 2, 1, 1, 1, 1, 1, 2, 1),
    ("mpd k5s1 inner7 C32->64 H90 B2",  32,  64,   90, 5, 1, 1, 2, 2, 0, 1, 2, 7),
    ("mpd k5s4 inner3 C32->64 H300 B2", 32,  64,  300, 5, 4, 1, 2, 2, 0, 1, 2, 3),
    ("descript k5s1 inner11 C1024 H74 B2", 1024, 1024,  74, 5, 1, 1, 2, 2, 0, 0, 2, 11),
    ("descript k5s3 inner11 C512->1024 H221 B2", 512, 1024, 221, 5, 3, 1, 2, 2, 0, 0, 2, 11),
    ("descript k5s3 inner2 C128->512 H3641 B2", 128, 512, 3641, 5, 3, 1, 2, 2, 0, 0, 2, 2),
]
sel = os.environ.get("ONLY")
s = torch.cuda.current_stream().cuda_stream


def rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


def timed(f):
    for _ in range(2):
        L.check(f())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(NIT):
        L.check(f())
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / NIT


print("%-36s %8s | %7s %7s %7s %8s | %7s %7s %7s %8s" % ("layer", "GFLOP", "f32 us", "x6 us", "x6 TF", "fwd err", "f32 us", "x6 us", "x6 TF", "dgr err"))
worst = 0.0
tot = [0.0, 0.0, 0.0, 0.0, 0.0]
torch.manual_seed(0)
for (name, ci, co, lin, k, st, dil, pl, pr, tr, act, b, inner) in layers:
    if sel and sel not in name:
        continue
    g = ConvGeom(stride=st, dilation=dil, pad_left=pl, pad_right=pr, transposed=bool(tr), act=act, slope=0.2, inner=inner)
    lout = g.out_len(lin, k)
    d = _desc(g, b, ci, co, lin, lout, k)
    r = C.byref(d)
    xs = (b, ci, lin, inner) if inner > 1 else (b, ci, lin)
    ys = (b, co, lout, inner) if inner > 1 else (b, co, lout)
    x = torch.randn(xs, device=dev)
    w = torch.randn((ci, co, k) if tr else (co, ci, k), device=dev) * 0.05
    bias = torch.randn(co, device=dev)
    res = torch.randn(ys, device=dev)
    dy = torch.randn(ys, device=dev)
    addg = torch.randn(xs, device=dev)
    wpf = torch.empty(L.lib.rh_conv1d_packed_floats(r, 0), device=dev)
    wpb = torch.empty(L.lib.rh_conv1d_packed_floats(r, 1), device=dev)
    L.check(L.lib.rh_conv1d_pack_f32(r, L.ptr(w), L.ptr(wpf), L.ptr(wpb), s))
    out = {}
    tim = {}
    # range slots (f16 build): the inputs' from rh_amax_f32, the outputs' are left by the launch and checked below
    RW = L.lib.rh_x6_range_words()
    use_r = L.lib.rh_x6_uses_ranges() == 1
    slots = torch.zeros(4, RW, device=dev, dtype=torch.int32)
    if use_r:
        L.check(L.lib.rh_amax_f32(L.ptr(x), x.numel(), L.ptr(slots[0]), s))
        L.check(L.lib.rh_amax_f32(L.ptr(dy), dy.numel(), L.ptr(slots[1]), s))
    for mode in ("0", "1"):
        os.environ["RH_CONV_X6"] = mode
        nf = L.lib.rh_conv1d_fwd_workspace_bytes(r); nd = L.lib.rh_conv1d_bwd_data_workspace_bytes(r)
        wsf = torch.empty(max(nf, 4) // 4, device=dev); wsd = torch.empty(max(nd, 4) // 4, device=dev)
        y = torch.full(ys, float("nan"), device=dev)
        dx = torch.full(xs, float("nan"), device=dev)
        arm = use_r and mode == "1"
        ff = lambda: (arm and L.lib.rh_x6_set_ranges(None, L.ptr(slots[0]), L.ptr(slots[2]), None)) or L.lib.rh_conv1d_fwd_f32(r, L.ptr(x), L.ptr(wpf), L.ptr(bias), None, L.ptr(res), L.ptr(y), L.ptr(wsf), nf, s)
        fd = lambda: (arm and L.lib.rh_x6_set_ranges(None, L.ptr(slots[1]), L.ptr(slots[3]), None)) or L.lib.rh_conv1d_bwd_data_f32(r, L.ptr(dy), L.ptr(wpb), L.ptr(x), None, L.ptr(addg), L.ptr(dx), L.ptr(wsd), nd, s)
        tim[mode] = (timed(ff), timed(fd))
        out[mode] = (y.clone(), dx.clone())
    if use_r:      # the published output ranges must cover the outputs (and not by much more than a bias)
        for nm, t_, sl in (("y", out["1"][0], slots[2]), ("dx", out["1"][1], slots[3])):
            pub = float(sl.view(torch.float32).max())
            true = float(t_.abs().max())
            if not (pub >= true and pub <= true + 8.0):
                print(f"   RANGE SLOT WRONG for {nm}: published {pub} true {true}")
                worst = float("inf")
    ef, ed = rel(out["1"][0], out["0"][0]), rel(out["1"][1], out["0"][1])
    if not (ef == ef and ed == ed):
        ef = ed = float("inf")
    worst = max(worst, ef, ed)
    flop = 2.0 * b * co * ci * k * (lin if tr else lout) * inner
    tot[0] += flop; tot[1] += tim["0"][0]; tot[2] += tim["1"][0]; tot[3] += tim["0"][1]; tot[4] += tim["1"][1]
    print("%-36s %8.2f | %7.1f %7.1f %7.1f %8.1e | %7.1f %7.1f %7.1f %8.1e" % (
        name, flop / 1e9, tim["0"][0], tim["1"][0], flop / tim["1"][0] / 1e6, ef, tim["0"][1], tim["1"][1], flop / tim["1"][1] / 1e6, ed))
print("TOTAL %38.2f | %7.1f %7.1f %7.1f          | %7.1f %7.1f %7.1f" % (tot[0] / 1e9, tot[1], tot[2], tot[0] / tot[2] / 1e6, tot[3], tot[4], tot[0] / tot[4] / 1e6))
print("worst relative L2 difference x6 vs f32: %.2e" % worst)
sys.exit(0 if worst < 5e-6 else 1)
