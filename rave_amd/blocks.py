"""Drop-in mirrors of the reference's hot-path blocks (rave/blocks.py) on the HIP kernels.

Constructor signatures, module trees and therefore ``state_dict`` keys/shapes equal the
reference's (e.g. ``net.1.aligned.branches.0.net.1.weight_v``), so checkpoints are
interchangeable.  Execution is fused: every ``activation -> conv`` pair becomes one kernel call
(activation applied while staging the input tile), ``Residual(DilatedUnit)`` is a single
autograd node with the residual add in the second conv's epilogue, and weight norm runs in its
own HIP kernel.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence, Union

import numpy as np
import torch
import torch.nn as nn

from . import cc, ops
from .ops import ACT_LEAKY, ACT_NONE, ACT_SNAKE

_NORMALIZATION_MODE = "weight_norm"  # configs/v1.gin:41 ``blocks.normalization.mode = 'weight_norm'``


def set_normalization_mode(mode: str) -> None:
    global _NORMALIZATION_MODE
    if mode not in ("identity", "weight_norm"):
        raise Exception(f"Normalization mode {mode} not supported")
    _NORMALIZATION_MODE = mode


def weight_norm(module: nn.Module, name: str = "weight", dim: int = 0) -> nn.Module:
    """Same parametrisation and parameter names as ``torch.nn.utils.weight_norm`` (``weight_g`` with
    shape (rows,1,..), ``weight_v``); the product g v/||v|| is computed by rh_weight_norm_fwd_f32
    inside the conv modules' forward (cc._effective_weight)."""
    if dim != 0:
        raise NotImplementedError("rave_amd weight_norm: dim must be 0 (what the reference uses)")
    w = getattr(module, name)
    del module._parameters[name]
    dims = tuple(range(1, w.dim()))
    g = w.detach().pow(2).sum(dims, keepdim=True).sqrt()
    module.register_parameter(name + "_g", nn.Parameter(g))
    module.register_parameter(name + "_v", nn.Parameter(w.detach().clone()))
    return module


def remove_weight_norm(module: nn.Module, name: str = "weight") -> nn.Module:
    g = module._parameters.pop(name + "_g")
    v = module._parameters.pop(name + "_v")
    module.register_parameter(name, nn.Parameter(torch._weight_norm(v.detach(), g.detach(), 0)))
    return module


def normalization(module: nn.Module, mode: Optional[str] = None):
    """rave/blocks.py:15-22."""
    mode = mode or _NORMALIZATION_MODE
    if mode == "identity":
        return module
    if mode == "weight_norm":
        return weight_norm(module)
    raise Exception(f"Normalization mode {mode} not supported")



def _set_warmed_up(mod, state: bool) -> None:
    """rave/blocks.py:736-738 re-creates the ``warmed_up`` buffer from a host scalar on every training step (a
    host-to-device copy) and ``forward`` branches on it (a device-to-host sync).  Same
    buffer and state_dict here, but the value is shadowed on the host and the buffer is only rewritten -- in place --
    when it changes: no copy, no sync in the steady state, and the step can be recorded into a hipGraph."""
    state = bool(state)
    if getattr(mod, "_warmed_up_host", None) != state:
        mod._warmed_up_host = state
        mod.warmed_up.fill_(int(state))


def _forget_host_shadows(mod, _incompatible=None) -> None:
    mod.__dict__.pop("_host_flags", None)
    if "_warmed_up_host" in mod.__dict__:
        mod._warmed_up_host = None


def track_flag_buffers(mod) -> None:
    """Call from the constructor of a module that shadows flag buffers on the host (``warmed_up``, ``enabled``,
    ``inited``): a ``load_state_dict`` that reaches the module forgets the shadows, so the next forward reads the
    loaded buffers (the reference reads them on every forward)."""
    mod.register_load_state_dict_post_hook(_forget_host_shadows)


def host_flag(mod, name: str) -> bool:
    """Truth value of a flag BUFFER (``DiscreteEncoder.enabled``, ``EuclideanCodebook.inited``: tensors in the reference,
    tested with ``if tensor:`` = a device-to-host sync per forward).  Read from the device as the reference does, except
    while a hipGraph is being recorded, where the value seen by the preceding eager iterations is used."""
    buf = getattr(mod, name)
    cache = mod.__dict__.setdefault("_host_flags", {})
    if not (buf.is_cuda and torch.cuda.is_current_stream_capturing()) or name not in cache:
        cache[name] = bool(buf)
    return cache[name]


def _is_warmed_up(mod) -> bool:
    h = getattr(mod, "_warmed_up_host", None)
    if h is None:          # never set through set_warmed_up (e.g. right after load_state_dict): read the buffer once
        h = mod._warmed_up_host = bool(mod.warmed_up)
    return h


class Snake(nn.Module):
    """rave/blocks.py:852-860 on the HIP Snake kernels (forward, dx and the per-channel dalpha
    reduction).  Inference-only graphs may instead fuse it into the next conv (ACT_SNAKE)."""

    def __init__(self, dim: int) -> None:
        super().__init__()
        self.alpha = nn.Parameter(torch.ones(dim, 1))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.snake(x, self.alpha)


def _act_of(m: nn.Module):
    """(act code, slope, alpha) of an activation module, or None if ``m`` is not one."""
    if isinstance(m, nn.LeakyReLU):
        return ACT_LEAKY, float(m.negative_slope), None
    if isinstance(m, Snake):
        return ACT_SNAKE, 0.0, m.alpha
    return None


class AdaptiveInstanceNormalization(nn.Module):
    """rave/blocks.py:863-926.  Identity in training mode (:901-902).  In eval mode the module keeps running statistics of the
    maps it sees while ``learn_y`` / ``learn_x`` are set (per batch slot and channel: mean and unbiased std over time,
    ``update`` :876-879) and, once both sides hold statistics, transfers x onto the target statistics (``transfer`` :887-895);
    with the default buffers it returns x.  Statistics and the transfer run on the HIP kernels ``adain_stats_kernel`` /
    ``adain_transfer_kernel`` (misc.hip); the flag buffers are read on the host, as the reference's ``if self.learn_y`` does."""

    def __init__(self, dim: int) -> None:
        super().__init__()
        for n in ("x", "y"):
            self.register_buffer(f"mean_{n}", torch.zeros(cc.MAX_BATCH_SIZE, dim, 1))
            self.register_buffer(f"std_{n}", torch.ones(cc.MAX_BATCH_SIZE, dim, 1))
            self.register_buffer(f"learn_{n}", torch.zeros(1))
            self.register_buffer(f"num_update_{n}", torch.zeros(1))

    def update(self, target: torch.Tensor, source: torch.Tensor, num_updates: torch.Tensor) -> None:
        """rave/blocks.py:876-879 (the reference's signature; ``forward`` uses the fused statistics kernel instead)."""
        bs = source.shape[0]
        target[:bs] += (source - target[:bs]) / (num_updates + 1)

    def reset_x(self):
        self.mean_x.zero_()
        self.std_x.zero_().add_(1)
        self.num_update_x.zero_()

    def reset_y(self):
        self.mean_y.zero_()
        self.std_y.zero_().add_(1)
        self.num_update_y.zero_()

    def transfer(self, x: torch.Tensor) -> torch.Tensor:
        return ops.adain_transfer(x, self.mean_x, self.std_x, self.mean_y, self.std_y)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.training:
            return x
        if bool(self.learn_y):
            ops.adain_stats_update(x, self.mean_y, self.std_y, self.num_update_y)
            self.num_update_y += 1
            return x
        if bool(self.learn_x):
            ops.adain_stats_update(x, self.mean_x, self.std_x, self.num_update_x)
            self.num_update_x += 1
        if bool(self.num_update_x) and bool(self.num_update_y):
            x = self.transfer(x)
        return x


_DILATED_UNIT_ACTIVATION: Optional[Callable[[int], nn.Module]] = None


def set_dilated_unit_activation(activation: Optional[Callable[[int], nn.Module]]) -> None:
    """What ``blocks.DilatedUnit.activation = @blocks.Snake`` (configs/snake.gin) does through gin: the
    reference's EncoderV2 / GeneratorV2 do NOT forward their own ``activation`` to the units
    (rave/blocks.py:553-559, 662-669), the units have their own binding."""
    global _DILATED_UNIT_ACTIVATION
    _DILATED_UNIT_ACTIVATION = activation


class DilatedUnit(nn.Module):
    """rave/blocks.py:83-112: act -> WN(Conv k dil d) -> act -> WN(Conv 1x1)."""

    def __init__(self, dim: int, kernel_size: int, dilation: int,
                 activation: Optional[Callable[[int], nn.Module]] = None) -> None:
        super().__init__()
        if activation is None:
            activation = _DILATED_UNIT_ACTIVATION or (lambda dim: nn.LeakyReLU(.2))
        net = [
            activation(dim),
            normalization(cc.Conv1d(dim, dim, kernel_size=kernel_size, dilation=dilation, bias=False,
                                    padding=cc.get_padding(kernel_size, dilation=dilation))),
            activation(dim),
            normalization(cc.Conv1d(dim, dim, kernel_size=1, bias=False)),
        ]
        self.net = cc.CachedSequential(*net)
        self.cumulative_delay = net[1].cumulative_delay

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return run_fused(self.net, x)


class Residual(nn.Module):
    """rave/blocks.py:31-45.  When the wrapped module is a DilatedUnit the whole
    ``x + Conv1(act(Conv3(act(x))))`` runs as one fused autograd node."""

    def __init__(self, module, cumulative_delay=0):
        super().__init__()
        additional_delay = module.cumulative_delay
        self.aligned = cc.AlignBranches(module, nn.Identity(), delays=[additional_delay, 0])
        self.cumulative_delay = additional_delay + cumulative_delay

    def forward(self, x):
        unit = self.aligned.branches[0]
        if isinstance(unit, DilatedUnit):
            a0, c3, a2, c1 = unit.net[0], unit.net[1], unit.net[2], unit.net[3]
            f0, f2 = _act_of(a0), _act_of(a2)
            if f0 is not None and f2 is not None and f0[2] is None and f2[2] is None:
                w3, g3 = cc._wn_pair(c3)
                w1, g1 = cc._wn_pair(c1)
                return ops.residual_unit(x, w3, w1, c3.geom(f0[0], f0[1]), c1.geom(f2[0], f2[1]), w3_g=g3, w1_g=g1,
                                         pre3=getattr(c3, "_prepacked", None), pre1=getattr(c1, "_prepacked", None))
            # Snake: alpha needs its own gradient -> unfused activation, fused residual add
            h = c3(a0(x))
            return c1(a2(h), residual=x)
        if isinstance(unit, nn.Sequential) and len(unit) >= 2 and isinstance(unit[-1], cc.Conv1d) \
                and unit[-1].groups == 1:
            # v1 ResidualLayer (rave/blocks.py:48-80): [act, conv]* with the skip added in the last
            # conv's epilogue
            mods = list(unit)
            h = run_fused(nn.Sequential(*mods[:-2]), x) if len(mods) > 2 else x
            f = _act_of(mods[-2])
            if f is not None and f[2] is None:
                return mods[-1](h, act=f[0], slope=f[1], residual=x)
            return mods[-1](mods[-2](h), residual=x)
        x_net, x_res = self.aligned(x)
        return x_net + x_res


_CONV_TYPES = (cc.Conv1d, cc.ConvTranspose1d)


def run_fused(net: nn.Sequential, x: torch.Tensor) -> torch.Tensor:
    """Executes a CachedSequential fusing ``activation -> conv`` pairs into single kernel calls."""
    mods = list(net)
    i = 0
    while i < len(mods):
        m = mods[i]
        f = _act_of(m)
        if f is not None and f[2] is None and i + 1 < len(mods) and isinstance(mods[i + 1], _CONV_TYPES):
            x = mods[i + 1](x, act=f[0], slope=f[1])
            i += 2
            continue
        x = m(x)
        i += 1
    return x


def normalize_dilations(dilations: Union[Sequence[int], Sequence[Sequence[int]]], ratios: Sequence[int]):
    if isinstance(dilations[0], int):
        dilations = [dilations for _ in ratios]
    return dilations


def mod_sigmoid(x):
    """rave/core.py:23-24."""
    return 2 * torch.sigmoid(x) ** 2.3 + 1e-7


def amp_to_impulse_response(amp, target_size):
    """rave/core.py:48-70 (irfft / roll / hann / pad / roll: stock torch + rocFFT, not a conv)."""
    amp = torch.stack([amp, torch.zeros_like(amp)], -1)
    amp = torch.view_as_complex(amp)
    amp = torch.fft.irfft(amp)
    filter_size = amp.shape[-1]
    amp = torch.roll(amp, filter_size // 2, -1)
    win = torch.hann_window(filter_size, dtype=amp.dtype, device=amp.device)
    amp = amp * win
    amp = nn.functional.pad(amp, (0, int(target_size) - int(filter_size)))
    amp = torch.roll(amp, -filter_size // 2, -1)
    return amp


def fft_convolve(signal, kernel):
    """rave/core.py:72-82."""
    signal = nn.functional.pad(signal, (0, signal.shape[-1]))
    kernel = nn.functional.pad(kernel, (kernel.shape[-1], 0))
    output = torch.fft.irfft(torch.fft.rfft(signal) * torch.fft.rfft(kernel))
    return output[..., output.shape[-1] // 2:]


class NoiseGeneratorV2(nn.Module):
    """rave/blocks.py:243-292 (configs/v2_small.gin:42-46).  The three strided convolutions
    (kernel 2r, stride r, pad (r,0), no weight norm) run on the HIP kernels with the LeakyReLU between
    them fused; the filtered-noise synthesis (irfft / rfft) is FFT work on stock torch + rocFFT."""

    def __init__(self, in_size: int, hidden_size: int, data_size: int, ratios, noise_bands: int,
                 n_channels: int = 1, activation: Callable[[int], nn.Module] = lambda dim: nn.LeakyReLU(.2)):
        super().__init__()
        net = []
        self.n_channels = n_channels
        channels = [in_size]
        channels.extend((len(ratios) - 1) * [hidden_size])
        channels.append(data_size * noise_bands * n_channels)
        for i, r in enumerate(ratios):
            net.append(cc.Conv1d(channels[i], channels[i + 1], 2 * r, padding=(r, 0), stride=r, bias=False))
            if i != len(ratios) - 1:
                net.append(activation(channels[i + 1]))
        self.net = nn.Sequential(*net)
        self.data_size = data_size
        self.register_buffer("target_size", torch.tensor(int(np.prod(ratios))).long())

    def forward(self, x, act: int = ACT_NONE, slope: float = 0.2, noise: Optional[torch.Tensor] = None):
        """``act``: activation of the CALLER fused into the first conv; ``noise`` injects the
        U(-1,1) draw (parity runs)."""
        mods = list(self.net)
        h = mods[0](x, act=act, slope=slope)
        h = run_fused(nn.Sequential(*mods[1:]), h) if len(mods) > 1 else h
        amp = mod_sigmoid(h - 5)
        amp = amp.permute(0, 2, 1)
        amp = amp.reshape(amp.shape[0], amp.shape[1], self.n_channels * self.data_size, -1)
        ir = amp_to_impulse_response(amp, self.target_size)
        if noise is None:
            noise = torch.rand_like(ir) * 2 - 1
        out = fft_convolve(noise, ir).permute(0, 2, 1, 3)
        return out.reshape(out.shape[0], out.shape[1], -1)


class EncoderV2(nn.Module):
    """rave/blocks.py:514-596."""

    def __init__(self, data_size: Union[int, None], capacity: int, ratios: Sequence[int], latent_size: int,
                 n_out: int, kernel_size: int, dilations: Sequence[int], keep_dim: bool = False,
                 recurrent_layer: Optional[Callable[[], nn.Module]] = None, n_channels: int = 1,
                 activation: Callable[[int], nn.Module] = lambda dim: nn.LeakyReLU(.2),
                 adain: Optional[Callable[[int], nn.Module]] = None, spectrogram=None) -> None:
        super().__init__()
        dilations_list = normalize_dilations(dilations, ratios)
        data_size = data_size or n_channels
        net = [normalization(cc.Conv1d(data_size * n_channels, capacity, kernel_size=kernel_size * 2 + 1,
                                       padding=cc.get_padding(kernel_size * 2 + 1), bias=False))]
        num_channels = capacity
        for r, dils in zip(ratios, dilations_list):
            for d in dils:
                if adain is not None:
                    net.append(adain(dim=num_channels))
                net.append(Residual(DilatedUnit(dim=num_channels, kernel_size=kernel_size, dilation=d)))
            net.append(activation(num_channels))
            out_channels = num_channels * r if keep_dim else num_channels * 2
            net.append(normalization(cc.Conv1d(num_channels, out_channels, kernel_size=2 * r, stride=r,
                                               padding=cc.get_padding(2 * r, r), bias=False)))
            num_channels = out_channels
        net.append(activation(num_channels))
        net.append(normalization(cc.Conv1d(num_channels, latent_size * n_out, kernel_size=kernel_size,
                                           padding=cc.get_padding(kernel_size), bias=False)))
        if recurrent_layer is not None:
            net.append(recurrent_layer(latent_size * n_out))
        self.net = cc.CachedSequential(*net)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return run_fused(self.net, x)


class GeneratorV2(nn.Module):
    """rave/blocks.py:599-714 (noise_module=None path; NoiseGeneratorV2 is FFT work outside the
    conv hot path and is not mirrored yet)."""

    def __init__(self, capacity: int, ratios: Sequence[int], latent_size: int, kernel_size: int,
                 dilations: Sequence[int], keep_dim: bool = False, data_size: Union[int, None] = None,
                 recurrent_layer: Optional[Callable[[], nn.Module]] = None, n_channels: int = 1,
                 amplitude_modulation: bool = False, noise_module=None,
                 activation: Callable[[int], nn.Module] = lambda dim: nn.LeakyReLU(.2),
                 adain: Optional[Callable[[int], nn.Module]] = None) -> None:
        super().__init__()
        data_size = n_channels if data_size is None else data_size * n_channels
        dilations_list = normalize_dilations(dilations, ratios)[::-1]
        ratios = ratios[::-1]
        num_channels = int(np.prod(ratios)) * capacity if keep_dim else 2 ** len(ratios) * capacity
        net = []
        if recurrent_layer is not None:
            net.append(recurrent_layer(latent_size))
        net.append(normalization(cc.Conv1d(latent_size, num_channels, kernel_size=kernel_size,
                                           padding=cc.get_padding(kernel_size), bias=False)))
        for r, dils in zip(ratios, dilations_list):
            out_channels = num_channels // r if keep_dim else num_channels // 2
            net.append(activation(num_channels))
            net.append(normalization(cc.ConvTranspose1d(num_channels, out_channels, 2 * r, stride=r,
                                                        padding=r // 2, bias=False)))
            num_channels = out_channels
            for d in dils:
                if adain is not None:
                    net.append(adain(num_channels))
                net.append(Residual(DilatedUnit(dim=num_channels, kernel_size=kernel_size, dilation=d)))
        net.append(activation(num_channels))
        waveform_module = normalization(cc.Conv1d(num_channels, data_size * 2 if amplitude_modulation else data_size,
                                                  kernel_size=kernel_size * 2 + 1,
                                                  padding=cc.get_padding(kernel_size * 2 + 1), bias=False))
        self.noise_module = None
        self.waveform_module = None
        if noise_module is not None:
            self.waveform_module = waveform_module
            self.noise_module = noise_module(out_channels, n_channels=n_channels)
        else:
            net.append(waveform_module)
        self.net = cc.CachedSequential(*net)
        self.amplitude_modulation = amplitude_modulation

    def forward(self, x: torch.Tensor, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        if self.noise_module is None:
            x = run_fused(self.net, x)
            if self.amplitude_modulation:
                return ops.amp_tanh(x)          # tanh(a * sigmoid(m)), rave/blocks.py:705-711
            return torch.tanh(x)
        # rave/blocks.py:696-711 with a noise branch: the last activation of ``net`` feeds BOTH the
        # noise generator and the waveform conv -> fuse it into both convs instead of materialising it
        mods = list(self.net)
        f = _act_of(mods[-1])
        if f is not None and f[2] is None:
            h = run_fused(nn.Sequential(*mods[:-1]), x)
            n = self.noise_module(h, act=f[0], slope=f[1], noise=noise)
            x = self.waveform_module(h, act=f[0], slope=f[1])
        else:
            h = run_fused(self.net, x)
            n = self.noise_module(h, noise=noise)
            x = self.waveform_module(h)
        if self.amplitude_modulation:
            x, amplitude = x.split(x.shape[1] // 2, 1)
            x = x * torch.sigmoid(amplitude)
        return torch.tanh(x + n)

    def set_warmed_up(self, state: bool):
        pass


def _reparam_fused() -> bool:
    """Reparametrisation + KL on one HIP kernel pair instead of ~30 elementwise ATen launches (-0.16 ms per v2 step);
    ``RH_REPARAM_FUSED=0`` restores the ATen chain.  Its latents differ from the chain's in the last bit (softplus / fused
    multiply-add): within every tolerance, but it can flip LeakyReLU gates of near-zero pre-activations downstream -- round
    3 kept it opt-in for that reason; round 4 counts the flips
    (tests/test_gpu_parity.py::test_v2_full_width_with_the_fused_reparametrisation_gate_flips_counted) and made it the default."""
    import os
    return os.environ.get("RH_REPARAM_FUSED", "1") != "0"


class VariationalEncoder(nn.Module):
    """rave/blocks.py:717-745 (elementwise latent maths on a (B, 2*latent, 32) tensor: torch ops)."""

    def __init__(self, encoder, beta: float = 1.0, n_channels=1):
        super().__init__()
        self.encoder = encoder(n_channels=n_channels)
        self.beta = beta
        self.register_buffer("warmed_up", torch.tensor(0))
        track_flag_buffers(self)

    def reparametrize(self, z, eps: Optional[torch.Tensor] = None):
        if z.is_cuda and z.dim() == 3 and z.dtype == torch.float32 and z.shape[1] % 2 == 0 and _reparam_fused():
            # the same arithmetic in two HIP launches (+ one for the gradient) instead of ~30 elementwise ATen kernels
            from . import ops
            b, c2, l = z.shape
            noise = torch.randn(b, c2 // 2, l, device=z.device, dtype=z.dtype) if eps is None else eps
            zs, kl = ops.reparametrize(z, noise)
            return zs, self.beta * kl
        mean, scale = z.chunk(2, 1)
        std = nn.functional.softplus(scale) + 1e-4
        var = std * std
        logvar = torch.log(var)
        noise = torch.randn_like(mean) if eps is None else eps
        z = noise * std + mean
        kl = (mean * mean + var - logvar - 1).sum(1).mean()
        return z, self.beta * kl

    def set_warmed_up(self, state: bool):
        _set_warmed_up(self, state)

    def forward(self, x: torch.Tensor):
        z = self.encoder(x)
        if _is_warmed_up(self):
            z = z.detach()
        return z


def _noise_like_reference(z: torch.Tensor, channels: int) -> torch.Tensor:
    """The noise-augmentation draw of rave/blocks.py:783-786 / :820-823: ``torch.randn(...).type_as(z)``, i.e. the CPU
    generator + a host-to-device copy, kept as is for eager steps (same random stream as the reference).  While a hipGraph is
    being recorded a pageable host-to-device copy is not capturable: the draw then comes from the device generator (torch
    advances its Philox offset per replay)."""
    if z.is_cuda and torch.cuda.is_current_stream_capturing():
        return torch.randn(z.shape[0], channels, z.shape[-1], device=z.device, dtype=z.dtype)
    return torch.randn(z.shape[0], channels, z.shape[-1]).type_as(z)


class WasserteinEncoder(nn.Module):
    """rave/blocks.py:748-791 (configs/wasserstein.gin): MMD regulariser on the latent -- elementwise /
    small-matrix torch ops around the HIP encoder; ``eps`` arguments inject the random draws."""

    def __init__(self, encoder_cls, noise_augmentation: int = 0, n_channels: int = 1):
        super().__init__()
        self.encoder = encoder_cls(n_channels=n_channels)
        self.register_buffer("warmed_up", torch.tensor(0))
        track_flag_buffers(self)
        self.noise_augmentation = noise_augmentation

    def compute_mean_kernel(self, x, y):
        kernel_input = (x[:, None] - y[None]).pow(2).mean(2) / x.shape[-1]
        return torch.exp(-kernel_input).mean()

    def compute_mmd(self, x, y):
        return self.compute_mean_kernel(x, x) + self.compute_mean_kernel(y, y) - 2 * self.compute_mean_kernel(x, y)

    def reparametrize(self, z, eps: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None):
        z_reshaped = z.permute(0, 2, 1).reshape(-1, z.shape[1])
        reg = self.compute_mmd(z_reshaped, torch.randn_like(z_reshaped) if eps is None else eps)
        if self.noise_augmentation:
            if noise is None:
                noise = _noise_like_reference(z, self.noise_augmentation)
            z = torch.cat([z, noise], 1)
        return z, reg.mean()

    def set_warmed_up(self, state: bool):
        _set_warmed_up(self, state)

    def forward(self, x: torch.Tensor):
        z = self.encoder(x)
        if _is_warmed_up(self):
            z = z.detach()
        return z


class SphericalEncoder(nn.Module):
    """rave/blocks.py:833-849 (configs/spherical.gin)."""

    def __init__(self, encoder_cls, n_channels: int = 1) -> None:
        super().__init__()
        self.encoder = encoder_cls(n_channels=n_channels)

    def reparametrize(self, z, eps=None):
        norm_z = z / torch.norm(z, p=2, dim=1, keepdim=True)
        return norm_z, torch.zeros_like(z).mean()

    def set_warmed_up(self, state: bool):
        pass

    def forward(self, x: torch.Tensor):
        return self.encoder(x)


class DiscreteEncoder(nn.Module):
    """rave/blocks.py:794-830 (configs/discrete.gin).  ``enabled`` == 0 (the state the shipped trainer
    always leaves it in: QuantizeCallback.on_train_batch_ is never called by Lightning -- SURVEY.md
    Appendix B #10): pass-through + ``noise_augmentation`` extra noise channels.  ``enabled`` != 0: the
    residual vector quantiser ``vq_cls`` (rave_amd.quantization.ResidualVectorQuantization on the HIP VQ
    kernels) replaces z by its quantisation and returns the commitment loss."""

    def __init__(self, encoder_cls, vq_cls=None, num_quantizers: int = 16, noise_augmentation: int = 0,
                 n_channels: int = 1):
        super().__init__()
        self.encoder = encoder_cls(n_channels=n_channels)
        self.rvq = vq_cls() if vq_cls is not None else None
        self.num_quantizers = num_quantizers
        self.register_buffer("warmed_up", torch.tensor(0))
        track_flag_buffers(self)
        self.register_buffer("enabled", torch.tensor(0))
        self.noise_augmentation = noise_augmentation

    def reparametrize(self, z, eps=None, noise: Optional[torch.Tensor] = None):
        """``eps`` / ``noise``: inject the noise-augmentation draw (parity runs); the step passes it as ``eps``."""
        if noise is None:
            noise = eps
        if host_flag(self, "enabled"):
            if self.rvq is None:
                raise RuntimeError("rave_amd DiscreteEncoder: enabled but constructed without vq_cls")
            z, diff, _ = self.rvq(z)
        else:
            diff = torch.zeros_like(z).mean()
        if self.noise_augmentation:
            if noise is None:
                noise = _noise_like_reference(z, self.noise_augmentation)
            z = torch.cat([z, noise], 1)
        return z, diff

    def set_warmed_up(self, state: bool):
        _set_warmed_up(self, state)

    def forward(self, x):
        return self.encoder(x)


# --------------------------------------------------------------------------------------------- v1
class SampleNorm(nn.Module):
    """rave/blocks.py:25-28."""

    def forward(self, x):
        return x / torch.norm(x, 2, 1, keepdim=True)


class ResidualLayer(nn.Module):
    """rave/blocks.py:48-80."""

    def __init__(self, dim, kernel_size, dilations, cumulative_delay=0,
                 activation: Callable[[int], nn.Module] = lambda dim: nn.LeakyReLU(.2)):
        super().__init__()
        net = []
        for d in dilations:
            net.append(activation(dim))
            net.append(normalization(cc.Conv1d(dim, dim, kernel_size, dilation=d, bias=False,
                                               padding=cc.get_padding(kernel_size, dilation=d))))
        self.net = Residual(cc.CachedSequential(*net), cumulative_delay=cumulative_delay)
        self.cumulative_delay = self.net.cumulative_delay

    def forward(self, x):
        return self.net(x)


class ResidualBlock(nn.Module):
    """rave/blocks.py:115-143."""

    def __init__(self, dim, kernel_size, dilations_list, cumulative_delay=0) -> None:
        super().__init__()
        layers = [ResidualLayer(dim, kernel_size, dilations) for dilations in dilations_list]
        self.net = cc.CachedSequential(*layers, cumulative_delay=cumulative_delay)
        self.cumulative_delay = self.net.cumulative_delay

    def forward(self, x):
        return self.net(x)


class ResidualStack(nn.Module):
    """rave/blocks.py:146-164 (configs/v1.gin:67-69: kernel_sizes [3], dilations [[1,1],[3,1],[5,1]])."""

    def __init__(self, dim, kernel_sizes=(3,), dilations_list=((1, 1), (3, 1), (5, 1)), cumulative_delay=0) -> None:
        super().__init__()
        blocks = [ResidualBlock(dim, k, dilations_list) for k in kernel_sizes]
        self.net = cc.AlignBranches(*blocks, cumulative_delay=cumulative_delay)
        self.cumulative_delay = self.net.cumulative_delay

    def forward(self, x):
        outs = self.net(x)
        y = outs[0]
        for o in outs[1:]:
            y = y + o
        return y


class UpsampleLayer(nn.Module):
    """rave/blocks.py:167-195."""

    def __init__(self, in_dim, out_dim, ratio, cumulative_delay=0,
                 activation: Callable[[int], nn.Module] = lambda dim: nn.LeakyReLU(.2)):
        super().__init__()
        net = [activation(in_dim)]
        if ratio > 1:
            net.append(normalization(cc.ConvTranspose1d(in_dim, out_dim, 2 * ratio, stride=ratio,
                                                        padding=ratio // 2, bias=False)))
        else:
            net.append(normalization(cc.Conv1d(in_dim, out_dim, 3, padding=cc.get_padding(3), bias=False)))
        self.net = cc.CachedSequential(*net)
        self.cumulative_delay = 0

    def forward(self, x):
        return run_fused(self.net, x)


class NoiseGenerator(nn.Module):
    """rave/blocks.py:198-240 (v1; configs/v1.gin:71-73: ratios [4,4,4], noise_bands 5)."""

    def __init__(self, in_size, data_size, ratios=(4, 4, 4), noise_bands=5):
        super().__init__()
        net = []
        channels = [in_size] * len(ratios) + [data_size * noise_bands]
        for i, r in enumerate(ratios):
            net.append(cc.Conv1d(channels[i], channels[i + 1], 3, padding=cc.get_padding(3, r), stride=r, bias=False))
            if i != len(ratios) - 1:
                net.append(nn.LeakyReLU(.2))
        self.net = cc.CachedSequential(*net)
        self.data_size = data_size
        self.cumulative_delay = 0
        self.register_buffer("target_size", torch.tensor(int(np.prod(ratios))).long())

    def forward(self, x, noise: Optional[torch.Tensor] = None):
        amp = mod_sigmoid(run_fused(self.net, x) - 5)
        amp = amp.permute(0, 2, 1)
        amp = amp.reshape(amp.shape[0], amp.shape[1], self.data_size, -1)
        ir = amp_to_impulse_response(amp, self.target_size)
        if noise is None:
            noise = torch.rand_like(ir) * 2 - 1
        out = fft_convolve(noise, ir).permute(0, 2, 1, 3)
        return out.reshape(out.shape[0], out.shape[1], -1)


class Generator(nn.Module):
    """rave/blocks.py:322-421 (v1 decoder): WN Conv k7 -> [UpsampleLayer, ResidualStack]* -> wave /
    loudness / noise branches."""

    def __init__(self, latent_size, capacity, data_size, ratios, loud_stride, use_noise, n_channels: int = 1,
                 recurrent_layer=None):
        super().__init__()
        if recurrent_layer is not None:
            raise NotImplementedError("rave_amd v1 Generator: recurrent_layer (hybrid.gin) is not on the hot path")
        net = [normalization(cc.Conv1d(latent_size, 2 ** len(ratios) * capacity, 7, padding=cc.get_padding(7), bias=False))]
        for i, r in enumerate(ratios):
            in_dim = 2 ** (len(ratios) - i) * capacity
            out_dim = 2 ** (len(ratios) - i - 1) * capacity
            net.append(UpsampleLayer(in_dim, out_dim, r))
            net.append(ResidualStack(out_dim))
        self.net = cc.CachedSequential(*net)
        wave_gen = normalization(cc.Conv1d(out_dim, data_size * n_channels, 7, padding=cc.get_padding(7), bias=False))
        loud_gen = normalization(cc.Conv1d(out_dim, 1, 2 * loud_stride + 1, stride=loud_stride, bias=False,
                                           padding=cc.get_padding(2 * loud_stride + 1, loud_stride)))
        branches = [wave_gen, loud_gen]
        if use_noise:
            branches.append(NoiseGenerator(out_dim, data_size * n_channels))
        self.synth = cc.AlignBranches(*branches)
        self.use_noise = use_noise
        self.loud_stride = loud_stride
        self.cumulative_delay = 0
        self.register_buffer("warmed_up", torch.tensor(0))
        track_flag_buffers(self)

    def set_warmed_up(self, state: bool):
        _set_warmed_up(self, state)

    def forward(self, x, noise: Optional[torch.Tensor] = None):
        x = self.net(x)
        br = self.synth.branches
        waveform, loudness = br[0](x), br[1](x)
        if self.loud_stride != 1:
            loudness = loudness.repeat_interleave(self.loud_stride)
        loudness = loudness.reshape(x.shape[0], 1, -1)
        waveform = torch.tanh(waveform) * mod_sigmoid(loudness)
        if _is_warmed_up(self) and self.use_noise:
            waveform = waveform + br[2](x, noise=noise)
        return waveform


class Encoder(nn.Module):
    """rave/blocks.py:424-503 (v1 encoder): Conv k7, [BatchNorm1d | SampleNorm, LeakyReLU, strided Conv]*,
    LeakyReLU, grouped Conv k5.  No weight norm; BatchNorm statistics are torch (not a conv)."""

    def __init__(self, data_size, capacity, latent_size, ratios, n_out, sample_norm, repeat_layers,
                 n_channels: int = 1, recurrent_layer=None, spectrogram=None):
        super().__init__()
        if recurrent_layer is not None:
            raise NotImplementedError("rave_amd v1 Encoder: recurrent_layer is not on the hot path")
        data_size = data_size or n_channels
        net = [cc.Conv1d(data_size * n_channels, capacity, 7, padding=cc.get_padding(7), bias=False)]
        for i, r in enumerate(ratios):
            in_dim = 2 ** i * capacity
            out_dim = 2 ** (i + 1) * capacity
            net.append(SampleNorm() if sample_norm else nn.BatchNorm1d(in_dim))
            net.append(nn.LeakyReLU(.2))
            net.append(cc.Conv1d(in_dim, out_dim, 2 * r + 1, padding=cc.get_padding(2 * r + 1, r), stride=r, bias=False))
            for _ in range(repeat_layers - 1):
                net.append(SampleNorm() if sample_norm else nn.BatchNorm1d(out_dim))
                net.append(nn.LeakyReLU(.2))
                net.append(cc.Conv1d(out_dim, out_dim, 3, padding=cc.get_padding(3), bias=False))
        net.append(nn.LeakyReLU(.2))
        net.append(cc.Conv1d(out_dim, latent_size * n_out, 5, padding=cc.get_padding(5), groups=n_out, bias=False))
        self.net = cc.CachedSequential(*net)
        self.cumulative_delay = 0

    def forward(self, x):
        return run_fused(self.net, x)
