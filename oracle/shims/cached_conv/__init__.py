"""TEST INFRASTRUCTURE ONLY -- oracle shim, never imported by the product path.

Restatement of the non-streaming semantics of the third-party package
``cached-conv>=2.5.0`` (reference requirements.txt:14), which is NOT vendored
under /root/reference and not installed in this image.  Semantics follow the
reference's own call sites: rave/pqmf.py:256-273,
rave/blocks.py:64-74,96-108,536-592,637-692, rave/discriminator.py:91-97 and
scripts/export_onnx.py:34-47 (``child._pad[0]``, ``cc.convs.Conv1d``).

Only the non-streaming mode (``use_cached_conv(False)``, the default and the
only mode used during training, scripts/train.py) is restated.
"""
import torch
import torch.nn as nn

MAX_BATCH_SIZE = 64
USE_BUFFER_CONV = False


def use_cached_conv(state: bool):
    global USE_BUFFER_CONV
    if state:
        raise NotImplementedError("oracle shim restates the non-streaming mode only")
    USE_BUFFER_CONV = False


def chunk_process(f, x, N):  # pragma: no cover - not on the hot path
    raise NotImplementedError


def get_padding(kernel_size, stride=1, dilation=1, mode="centered"):
    """'same' padding; ``stride`` is accepted and ignored (as upstream)."""
    if kernel_size == 1:
        return (0, 0)
    p = (kernel_size - 1) * dilation + 1
    if mode == "centered":
        p_right = p // 2
        p_left = (p - 1) // 2
    elif mode == "causal":
        p_right = 0
        p_left = p // 2 + (p - 1) // 2
    else:
        raise Exception(f"Padding mode {mode} is not valid")
    return (p_left, p_right)


class CachedPadding1d(nn.Module):  # streaming only; placeholder for isinstance checks
    def __init__(self, padding, crop=False):
        super().__init__()
        self.padding = padding
        self.crop = crop


class Conv1d(nn.Conv1d):
    def __init__(self, *args, **kwargs):
        self._pad = kwargs.get("padding", (0, 0))
        kwargs.pop("cumulative_delay", 0)
        kwargs["padding"] = 0
        super().__init__(*args, **kwargs)
        self.cumulative_delay = 0

    def script_cache(self):
        pass

    def forward(self, x):
        x = nn.functional.pad(x, self._pad)
        return nn.functional.conv1d(x, self.weight, self.bias, self.stride,
                                    self.padding, self.dilation, self.groups)


class ConvTranspose1d(nn.ConvTranspose1d):
    def __init__(self, *args, **kwargs):
        kwargs.pop("cumulative_delay", 0)
        super().__init__(*args, **kwargs)
        self.cumulative_delay = 0

    def script_cache(self):
        pass


class CachedSequential(nn.Sequential):
    def __init__(self, *args, **kwargs):
        cumulative_delay = kwargs.pop("cumulative_delay", 0)
        stride = kwargs.pop("stride", 1)
        super().__init__(*args, **kwargs)
        for m in reversed(list(args)):
            if hasattr(m, "cumulative_delay"):
                cumulative_delay = m.cumulative_delay
                break
        self.cumulative_delay = cumulative_delay
        self.stride = stride


class AlignBranches(nn.Module):
    def __init__(self, *branches, delays=None, cumulative_delay=0, stride=1):
        super().__init__()
        self.branches = nn.ModuleList(branches)
        self.cumulative_delay = cumulative_delay

    def forward(self, x):
        return [branch(x) for branch in self.branches]


class Branches(nn.Module):
    def __init__(self, *branches, delays=None, cumulative_delay=0, stride=1):
        super().__init__()
        self.branches = nn.ModuleList(branches)
        self.cumulative_delay = cumulative_delay

    def forward(self, x):
        return [branch(x) for branch in self.branches]


class _Convs:
    Conv1d = Conv1d
    ConvTranspose1d = ConvTranspose1d


convs = _Convs
