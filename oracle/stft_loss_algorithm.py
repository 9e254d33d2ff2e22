"""TEST INFRASTRUCTURE ONLY (not imported by the product): a numpy restatement of the ALGORITHM of rave_amd/csrc/stft_loss.hip
-- one complex transform per frame pair, Stockham radix-8 passes with the kernel's index arithmetic, the Hermitian-extended
gradient operand, the inverse transform as swap(FFT(swap(W))), overlap-add and the reflected-margin folds -- so that the
algebra the HIP kernel relies on is pinned on the CPU against torch.stft + autograd (the reference's formulation:
rave/core.py:269-344, torchaudio Spectrogram(normalized=True, power=None) -> abs -> relative L2 + L1 of logs)."""
import numpy as np

PLANS = {128: [8, 8, 2], 256: [8, 8, 4], 512: [8, 8, 8], 1024: [8, 8, 8, 2], 2048: [8, 8, 8, 4]}


def stockham_fft(z):
    """Forward DFT of a length-n complex vector with the kernel's pass structure: n/8 "threads" holding 8 points each;
    butterfly j = t + u n/8 of a radix-R pass takes in[j + r n/R], twiddles by exp(-2 pi i r k / (NS R)), k = j mod NS, and
    writes (j - k) R + k + r NS (stft_loss.hip: Fft<N>::load / store8)."""
    n = len(z)
    tw = np.exp(-2j * np.pi * np.arange(n) / n)
    tpf, buf, ns = n // 8, np.asarray(z, dtype=complex).copy(), 1
    for radix in PLANS[n]:
        out = np.zeros(n, complex)
        for t in range(tpf):
            for u in range(8 // radix):
                j = t + u * tpf
                k = j % ns
                v = np.array([buf[j + r * (n // radix)] * tw[r * k * (n // (ns * radix))] for r in range(radix)])
                spec = np.array([sum(v[r] * np.exp(-2j * np.pi * r * q / radix) for r in range(radix)) for q in range(radix)])
                for r in range(radix):
                    out[(j - k) * radix + k + r * ns] = spec[r]
        buf, ns = out, ns * radix
    return buf


def distance_and_gradients(x, y, win, eps, fft=np.fft.fft):
    """(distance, d/dx, d/dy) for rows x, y (R, T) at one scale n = len(win), hop n/4, the way the kernel computes them."""
    n = len(win)
    hop, (rows, t) = n // 4, x.shape
    nf = t // hop + 1
    refl = lambda p: np.where(p < 0, -p, np.where(p >= t, 2 * (t - 1) - p, p))
    idx = refl(np.arange(nf)[:, None] * hop + np.arange(n)[None, :] - n // 2)
    z = np.apply_along_axis(fft, -1, (x[:, idx] + 1j * y[:, idx]) * win)            # one transform per frame PAIR
    k = np.arange(n // 2 + 1)
    z1, z2 = z[..., k], z[..., (n - k) % n]
    sx, sy = (z1 + np.conj(z2)) / 2, (z1 - np.conj(z2)) / 2j
    a, b = np.abs(sx), np.abs(sy)
    sum_a, sum_b, cnt = ((a - b) ** 2).sum(), (a * a).sum(), a.size
    dist = sum_a / sum_b + np.abs(np.log(a + eps) - np.log(b + eps)).sum() / cnt
    d, sg = a - b, np.sign(a - b)                                                  # sign(log(a+eps) - log(b+eps)) = sign(a - b)
    da = 2 * d / sum_b - 2 * a * sum_a / sum_b ** 2 + sg / cnt / (a + eps)
    db = -2 * d / sum_b - sg / cnt / (b + eps)
    with np.errstate(invalid="ignore", divide="ignore"):
        gx = np.where(a > 0, sx * (da / a), 0)
        gy = np.where(b > 0, sy * (db / b), 0)
    w = np.zeros_like(z)
    inner = (k > 0) & (k < n // 2)
    w[..., k[inner]] = (gx[..., inner] + 1j * gy[..., inner]) / 2                  # Hermitian extension, half weights
    w[..., n - k[inner]] = (np.conj(gx[..., inner]) + 1j * np.conj(gy[..., inner])) / 2
    for kk in (0, n // 2):
        w[..., kk] = gx[..., kk].real + 1j * gy[..., kk].real
    swap = lambda c: c.imag + 1j * c.real
    wv = swap(np.apply_along_axis(fft, -1, swap(w)))                                # unnormalised inverse
    dxf, dyf = wv.real * win, wv.imag * win
    padx, pady = np.zeros((rows, t + n)), np.zeros((rows, t + n))
    for f in range(nf):                                                             # overlap-add in padded coordinates
        padx[:, f * hop:f * hop + n] += dxf[:, f]
        pady[:, f * hop:f * hop + n] += dyf[:, f]

    def fold(pad):                                                                  # adjoint of the reflect padding
        out = pad[:, n // 2:n // 2 + t].copy()
        for p in range(1, n // 2 + 1):
            out[:, p] += pad[:, n // 2 - p]
        for i in range(n // 2):
            out[:, t - 2 - i] += pad[:, t + n // 2 + i]
        return out
    return dist, fold(padx), fold(pady)
