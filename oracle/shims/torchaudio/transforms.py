import torch
import torch.nn as nn


class Spectrogram(nn.Module):
    def __init__(self, n_fft=400, win_length=None, hop_length=None, pad=0,
                 window_fn=torch.hann_window, power=2.0, normalized=False,
                 wkwargs=None, center=True, pad_mode="reflect", onesided=True, return_complex=None):
        super().__init__()
        self.n_fft = n_fft
        self.win_length = win_length if win_length is not None else n_fft
        self.hop_length = hop_length if hop_length is not None else self.win_length // 2
        window = window_fn(self.win_length) if wkwargs is None else window_fn(self.win_length, **wkwargs)
        self.register_buffer("window", window)
        self.pad = pad
        self.power = power
        self.normalized = normalized
        self.center = center
        self.pad_mode = pad_mode
        self.onesided = onesided

    def forward(self, x):
        if self.pad > 0:
            x = nn.functional.pad(x, (self.pad, self.pad))
        shape = x.shape
        x = x.reshape(-1, shape[-1])
        y = torch.stft(x, self.n_fft, self.hop_length, self.win_length,
                       window=self.window, center=self.center, pad_mode=self.pad_mode,
                       normalized=False, onesided=self.onesided, return_complex=True)
        y = y.reshape(shape[:-1] + y.shape[-2:])
        if self.normalized:
            y = y / self.window.pow(2.0).sum().sqrt()
        if self.power is not None:
            if self.power == 1.0:
                return y.abs()
            return y.abs().pow(self.power)
        return y


class MelSpectrogram(nn.Module):
    def __init__(self, *a, **k):
        raise NotImplementedError
