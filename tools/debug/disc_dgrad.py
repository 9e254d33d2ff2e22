"""Forward / data-gradient timings and launch plans of the strided discriminator convolutions (MSD k = 15 s = 4 at 64
waveforms, MPD k = 5 s = 4 period-major) -- why is the data gradient 2-3x slower than the forward?"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from rave_amd import _lib as L
from rave_amd.ops import ConvGeom, _desc
dev = torch.device("cuda:0")
# (name, B, cin, cout, L_in, k, stride, pad)
layers = [("msd k15s4 96->192", 64, 96, 192, 16384, 15, 4, 7), ("msd k15s4 192->384", 64, 192, 384, 4096, 15, 4, 7),
          ("msd k15s4 384->768", 64, 384, 768, 1024, 15, 4, 7),
          ("mpd k5s4 96->192 p2", 128, 96, 192, 8192, 5, 4, 2), ("mpd k5s4 192->384 p2", 128, 192, 384, 2048, 5, 4, 2),
          ("mpd k5s4 96->192 p11", 704, 96, 192, 1490, 5, 4, 2), ("mpd k5s4 192->384 p7", 448, 192, 384, 586, 5, 4, 2)]
s = torch.cuda.current_stream().cuda_stream
for (name, B, ci, co, lin, k, st, pad) in layers:
    for act in (1, 0):
        g = ConvGeom(stride=st, dilation=1, pad_left=pad, pad_right=pad, transposed=False, act=act, slope=0.2)
        lout = g.out_len(lin, k)
        d = _desc(g, B, ci, co, lin, lout, k)
        r = C.byref(d)
        x = torch.randn(B, ci, lin, device=dev); w = torch.randn(co, ci, k, device=dev) * 0.05
        y = torch.empty(B, co, lout, device=dev); dy = torch.randn(B, co, lout, device=dev); dx = torch.empty_like(x)
        wpf = torch.empty(L.lib.rh_conv1d_packed_floats(r, 0), device=dev); wpb = torch.empty(L.lib.rh_conv1d_packed_floats(r, 1), device=dev)
        L.check(L.lib.rh_conv1d_pack_f32(r, L.ptr(w), L.ptr(wpf), L.ptr(wpb), s))
        nf = L.lib.rh_conv1d_fwd_workspace_bytes(r); nd = L.lib.rh_conv1d_bwd_data_workspace_bytes(r)
        wsf = torch.empty(max(nf, 4) // 4, device=dev); wsd = torch.empty(max(nd, 4) // 4, device=dev)
        fns = [lambda: L.lib.rh_conv1d_fwd_f32(r, L.ptr(x), L.ptr(wpf), None, None, None, L.ptr(y), L.ptr(wsf), nf, s),
               lambda: L.lib.rh_conv1d_bwd_data_f32(r, L.ptr(dy), L.ptr(wpb), L.ptr(x), None, None, L.ptr(dx), L.ptr(wsd), nd, s)]
        res = []
        for f in fns:
            for _ in range(2): L.check(f())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): L.check(f())
            e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) * 1e3 / 5)
        flop = 2.0 * B * co * ci * k * lout
        plans = []
        for which in (0, 1):
            out = (C.c_int32 * 8)()
            L.lib.rh_conv1d_plan_info(r, which, 0, 0, out)
            plans.append(tuple(out))
        print("%-22s act %d  %7.1f GF | fwd %7.1f us %6.1f TF | dgrad %7.1f us %6.1f TF | plan fwd %s dgrad %s  (family,tm,tn,wm,ksplit,is,vs,wgs)"
              % (name, act, flop / 1e9, res[0], flop / res[0] / 1e6, res[1], flop / res[1] / 1e6, plans[0], plans[1]))
