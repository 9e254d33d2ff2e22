"""Adam for the generator / discriminator parameters on the HIP kernel of rave_amd/csrc/adam.hip (rh_adam_step_f32):
``torch.optim.Adam(params, lr, betas)`` semantics (weight_decay = 0, amsgrad = False -- what rave/model.py:226-233
configures), same ``state`` layout (``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter -- every parameter counts its
OWN steps, as torch's does: one that gets its first gradient late starts its bias corrections at 1 -- so ``state_dict()``
round-trips with torch's Adam), step counters and learning rate in device memory (always "capturable": the step records
into a hipGraph).  Plumbing beside the hot path: one pass over parameters, gradients and both moments at HBM speed instead of
torch's fused multi-tensor kernel at 1.6 TB/s."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        if not 0.0 <= float(eps) or not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError("FusedAdam: bad hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps))
        self._tables = {}
        self._gs = {}          # per group: device scratch (kept out of param_groups: state_dict() stays torch's)

    def _group_state(self, gi: int, group):
        ps = [p for p in group["params"] if p.requires_grad]
        if not ps:
            return None
        dev = ps[0].device
        gs = self._gs.setdefault(gi, {})
        if "dev" not in gs:
            gs["dev"] = dev
            gs["aux"] = None           # 2 floats per live parameter (bias corrections per step counter)
            gs["lr_dev"] = None
        return gs

    def load_state_dict(self, state_dict) -> None:
        """torch's loader, then every step counter as ONE f32 device scalar (a state_dict written by torch's Adam carries CPU /
        1-element / integer counters): nothing is left to convert inside ``step``, where a conversion would be a host
        synchronisation (illegal while a hipGraph records)."""
        super().load_state_dict(state_dict)
        for p, st in self.state.items():
            t = st.get("step")
            if t is not None and not (torch.is_tensor(t) and t.device == p.device and t.dtype == torch.float32 and t.dim() == 0):
                st["step"] = torch.as_tensor(float(t), dtype=torch.float32).to(p.device).reshape(())
        self._tables.clear()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            gs = self._group_state(gi, group)
            if gs is None:
                continue
            live = []
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("rave_amd FusedAdam: parameters must be contiguous fp32 tensors on the GPU")
                g = p.grad
                if g.is_sparse or g.dtype != torch.float32:
                    raise RuntimeError("rave_amd FusedAdam: dense fp32 gradients only")
                if not g.is_contiguous():
                    g = p.grad = g.contiguous()
                st = self.state[p]
                capturing = torch.cuda.is_current_stream_capturing()
                if not st:
                    if capturing:       # state created here would be re-zeroed by every replay of the recorded step
                        raise RuntimeError("rave_amd FusedAdam: a parameter received its first gradient inside a hipGraph capture; "
                                           "run one eager step (or GraphedTrainingStep's warm-up) first")
                    st["step"] = torch.zeros((), device=p.device, dtype=torch.float32)     # per parameter, as torch's Adam
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                else:
                    t = st["step"]
                    if not (torch.is_tensor(t) and t.device == p.device and t.dtype == torch.float32 and t.dim() == 0):
                        if capturing:   # float(t) is a host synchronisation: illegal while recording
                            raise RuntimeError("rave_amd FusedAdam: step counters of a loaded state_dict must be normalised by one "
                                               "eager step (or load_state_dict of this class) before a hipGraph capture")
                        # after load_state_dict of a foreign layout (CPU / int / 1-element counters): one device scalar
                        st["step"] = torch.as_tensor(float(t), dtype=torch.float32).to(p.device).reshape(())
                live.append((p, g, st["exp_avg"], st["exp_avg_sq"], st["step"]))
            if not live:
                continue
            lr = group["lr"]
            if torch.is_tensor(lr):
                lr_dev = lr
            else:                                               # a Python float: mirrored into a device scalar
                if gs["lr_dev"] is None:
                    gs["lr_dev"] = torch.empty((), device=gs["dev"], dtype=torch.float32)
                gs["lr_dev"].fill_(float(lr))
                lr_dev = gs["lr_dev"]
            key = tuple((p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), t.data_ptr()) for p, g, m, v, t in live)
            tab = self._tables.get(gi)
            if tab is None or tab[0] != key:
                arr = (L.AdamItem * len(live))()
                for i, (p, g, m, v, t) in enumerate(live):
                    arr[i].p, arr[i].g, arr[i].m, arr[i].v, arr[i].n = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()
                    arr[i].step = t.data_ptr()
                tab = self._tables[gi] = (key, arr)
            if gs["aux"] is None or gs["aux"].numel() < 2 * len(live):
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("rave_amd FusedAdam: more parameters received gradients inside a hipGraph capture than in "
                                       "the eager steps before it")
                gs["aux"] = torch.zeros(2 * len(live), device=gs["dev"], dtype=torch.float32)
            b1, b2 = group["betas"]
            L.check(L.lib.rh_adam_step_f32(tab[1], len(live), L.ptr(lr_dev), float(b1), float(b2), float(group["eps"]),
                                           L.ptr(gs["aux"]), L.stream()), "adam_step")
        return loss
