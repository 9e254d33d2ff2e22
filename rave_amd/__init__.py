"""rave_amd -- MI355X (gfx950) native hot path of RAVE: PQMF filterbank + Conv1d residual stacks.

``rave_amd.cc`` mirrors the ``cached_conv`` operator API the reference's blocks are written
against, ``rave_amd.pqmf`` / ``rave_amd.blocks`` / ``rave_amd.discriminator``
mirror the reference's leaf modules with identical constructor signatures and state_dict
layouts; all arithmetic runs in hand-written HIP kernels behind the C ABI of
``include/rave_hip.h`` (``rave_amd/librave_hip.so``).
"""
__version__ = "0.1.0"
