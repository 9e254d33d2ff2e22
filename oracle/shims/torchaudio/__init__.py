"""TEST INFRASTRUCTURE ONLY -- oracle shim for torchaudio (reference
requirements.txt:16): only ``transforms.Spectrogram`` (rave/core.py:286-292,
rave/discriminator.py:12-20) restated on ``torch.stft``.  Parity at this
boundary is unpinned (no reference test); it only feeds the non-hot-path
spectral losses."""
from . import transforms, functional  # noqa: F401
