"""Stand-in for ``rave.model.RAVE`` (rave/model.py:133-424) used where the reference package
cannot travel (the GPU box): same sub-module names (``pqmf``, ``encoder``, ``decoder``,
``discriminator`` -> identical state_dict keys for the hot path) and a ``training_step`` that
restates rave/model.py:288-413 statement by statement, driving the HIP drop-in modules.

With the reference installed nothing here is needed: ``rave.model.RAVE`` itself runs on the
drop-ins.  This file is host-side plumbing, not the product.
"""
from __future__ import annotations

from functools import partial
from typing import Callable, Dict, Optional

import torch
import torch.nn as nn

from . import blocks, cc, descript_discriminator, discriminator, losses, pqmf, quantization

_default_loss_weights = {
    "audio_distance": 1.,
    "multiband_audio_distance": 1.,
    "adversarial": 1.,
    "feature_matching": 20,
}


def _loss_combine_fused() -> bool:
    """RH_LOSS_COMBINE=0: the generator loss as the reference's chain of scalar ATen operations (same bits)."""
    import os
    return os.environ.get("RH_LOSS_COMBINE", "1") != "0"


def _pqmf_encode(pq, x: torch.Tensor):
    """rave/model.py:116-122."""
    batch_size = x.shape[:-2]
    x_multiband = x.reshape(-1, 1, x.shape[-1])
    x_multiband = pq(x_multiband)
    return x_multiband.reshape(*batch_size, -1, x_multiband.shape[-1])


def valid_signal_crop(x: torch.Tensor, left_rf, right_rf):
    """rave/core.py:220-225: drop the samples the receptive field does not fully cover (in band samples: the
    receptive field is measured in audio samples, ``dim`` = the multiband channel count)."""
    dim = x.shape[1]
    x = x[..., int(left_rf) // dim:]
    if int(right_rf):
        x = x[..., :-int(right_rf) // dim]
    return x


def _pqmf_decode(pq, x: torch.Tensor, batch_size, n_channels: int):
    """rave/model.py:125-130."""
    x = x.reshape(x.shape[0] * n_channels, -1, x.shape[-1])
    x = pq.inverse(x)
    return x.reshape(*batch_size, n_channels, -1)


class RAVE(nn.Module):
    def __init__(self, latent_size, sampling_rate, encoder, decoder, discriminator, phase_1_duration,
                 gan_loss, valid_signal_crop, feature_matching_fun, num_skipped_features,
                 audio_distance: Callable[[], nn.Module], multiband_audio_distance: Callable[[], nn.Module],
                 n_bands: int = 16, weights: Optional[Dict[str, float]] = None,
                 pqmf: Optional[Callable[[], nn.Module]] = None, update_discriminator_every: int = 2,
                 n_channels: int = 1):
        super().__init__()
        self.pqmf = pqmf(n_channels=n_channels)
        self.encoder = encoder(n_channels=n_channels)
        self.decoder = decoder(n_channels=n_channels)
        self.discriminator = discriminator(n_channels=n_channels)
        self.audio_distance = audio_distance()
        self.multiband_audio_distance = multiband_audio_distance()
        self.gan_loss = gan_loss
        self.latent_size = latent_size
        self.warmup = phase_1_duration
        self.weights = dict(_default_loss_weights)
        self.weights.update(weights or {})
        self.warmed_up = False
        self.sr = sampling_rate
        self.valid_signal_crop = valid_signal_crop
        self.n_channels = n_channels
        self.feature_matching_fun = feature_matching_fun
        self.num_skipped_features = num_skipped_features
        self.update_discriminator_every = update_discriminator_every
        self.beta_factor = 1.
        self._beta_dev = None          # 0-d device copy of beta_factor read by the recorded (hipGraph) step
        self.register_buffer("receptive_field", torch.tensor([0, 0]).long())
        self.logged: Dict[str, torch.Tensor] = {}
        self._opts = None
        self._gen_sched = None
        self._prep = None
        # OPT-IN (default off = the reference's exact work per step): skip gradient work
        # whose result the reference computes and then discards -- on generator steps the discriminator's weight
        # gradients (dis_opt.zero_grad() runs before they could be used), on discriminator steps everything behind
        # y_raw / x_raw (gen_opt.zero_grad() discards it).  Parameter trajectories are identical.
        self.skip_dead_grads = False

    def beta_device(self, device) -> torch.Tensor:
        """0-d device tensor holding ``beta_factor`` (created once, updated in place: see training_step)."""
        if self._beta_dev is None or self._beta_dev.device != device:
            self._beta_dev = torch.full((), float(self.beta_factor), device=device, dtype=torch.float32)
        elif not (device.type == "cuda" and torch.cuda.is_current_stream_capturing()):
            self._beta_dev.fill_(float(self.beta_factor))
        return self._beta_dev

    def prepare_weights(self, with_discriminator: bool = False, reuse: bool = False):
        """Refresh weight norm + packed weights of every conv in two launches (see rave_amd/prep.py);
        pair with release_weights().  training_step does this itself.  ``reuse=True``: inference forwards -- keep the
        packed operands while no parameter changed (rave_amd.prep.WeightPrep.run)."""
        from .prep import WeightPrep
        if self._prep is None:
            self._prep = (WeightPrep(nn.ModuleList([self.encoder, self.decoder])), WeightPrep(self.discriminator))
        self._prep[0].run(reuse)
        if with_discriminator:
            self._prep[1].run(reuse)

    def set_phase_flags_eagerly(self) -> None:
        """Read every buffer-derived host flag once outside a capture (so that a following capture finds them cached
        with the CURRENT buffer values) and push the warm-up state to the blocks."""
        from .blocks import host_flag
        self.encoder.set_warmed_up(self.warmed_up)
        self.decoder.set_warmed_up(self.warmed_up)
        for mod in self.modules():
            for name in ("enabled", "inited"):
                if torch.is_tensor(getattr(mod, name, None)):
                    host_flag(mod, name)

    def release_weights(self):
        if self._prep is not None:
            self._prep[0].release()
            self._prep[1].release()

    # ---- rave/model.py:226-236
    def configure_optimizers(self, capturable: bool = False):
        """``capturable``: step counters and learning rates live on the device, so that the optimizer step can be
        recorded into a hipGraph (GraphedTrainingStep); same update."""
        gen_p = list(self.encoder.parameters()) + list(self.decoder.parameters())
        dis_p = list(self.discriminator.parameters())
        # same optimiser and hyper-parameters as the reference; on the GPU PyTorch's single-kernel ("fused")
        # implementation of the same update replaces the default multi-pass foreach one
        fused = all(p.is_cuda for p in gen_p + dis_p)
        lr_g, lr_d = 1e-3, 1e-4
        kw = {}
        if capturable:
            dev = gen_p[0].device
            lr_g, lr_d = torch.tensor(lr_g, device=dev), torch.tensor(lr_d, device=dev)
            kw["capturable"] = True
        import os
        if fused and os.environ.get("RH_ADAM", "1") != "0":
            # the same update on one HIP kernel pass at HBM speed (rave_amd/optim.py; RH_ADAM=0: torch's fused kernel)
            from .optim import FusedAdam
            gen_opt = FusedAdam(gen_p, lr_g, (.5, .9))
            dis_opt = FusedAdam(dis_p, lr_d, (.5, .9))
        else:
            gen_opt = torch.optim.Adam(gen_p, lr_g, (.5, .9), fused=fused, **kw)
            dis_opt = torch.optim.Adam(dis_p, lr_d, (.5, .9), fused=fused, **kw)
        self._opts = (gen_opt, dis_opt)
        # rave/model.py:234-236: the generator learning rate decays linearly to 0.1x over phase 1
        self._gen_sched = LinearLR(gen_opt, start_factor=1.0, end_factor=0.1, total_iters=self.warmup)
        return gen_opt, dis_opt

    def lr_schedulers(self):
        self.optimizers()
        return self._gen_sched

    def on_train_batch_end(self, outputs=None, batch=None, batch_idx=None) -> None:
        """rave/model.py:272-274 (Lightning hook; a driver loop calls it after every training_step)."""
        self.lr_schedulers().step()

    def optimizers(self):
        if self._opts is None:
            self.configure_optimizers()
        return self._opts

    # ---- rave/model.py:244-270
    def encode(self, x, return_mb: bool = False):
        x_enc = _pqmf_encode(self.pqmf, x)
        z = self.encoder(x_enc)
        if return_mb:
            return z, x_enc
        return z

    def decode(self, z):
        batch_size = z.shape[:-2]
        y = self.decoder(z)
        return _pqmf_decode(self.pqmf, y, batch_size=batch_size, n_channels=self.n_channels)

    def forward(self, x):
        z = self.encode(x, return_mb=False)
        z = self.encoder.reparametrize(z)[0]
        return self.decode(z)

    def _fused_feature_matching(self, features):
        """The feature-matching distance of rave/model.py:359-372 in one HIP pass over the unsplit feature maps
        (rave_amd.ops.feature_matching) when the configured function is ``mean_difference(norm="L1"[, relative])`` on GPU
        tensors; None -> the generic per-map formulation runs.  ``RH_FM_FUSED=0`` disables."""
        import os
        fun = self.feature_matching_fun
        if os.environ.get("RH_FM_FUSED", "1") == "0" or not isinstance(fun, partial) or fun.func is not losses.mean_difference or fun.args:
            return None
        kw = dict(fun.keywords)
        if kw.pop("norm", "L1") != "L1":
            return None
        relative = bool(kw.pop("relative", False))
        if kw:
            return None
        maps, weights = [], []
        for scale in features:
            used = list(scale[self.num_skipped_features:])
            if not used:
                return None
            for f in used:
                if not (torch.is_tensor(f) and f.is_cuda and f.dtype == torch.float32 and f.shape[0] % 2 == 0 and f.numel() > 0):
                    return None
                maps.append(f)
                weights.append(1.0 / (len(used) * len(features)))
        if not maps:
            return None
        from . import ops
        total = None
        for i in range(0, len(maps), 80):          # one kernel-argument table holds 80 maps (descript: > 100 per step)
            part = ops.feature_matching(maps[i:i + 80], weights[i:i + 80], relative)
            total = part if total is None else total + part
        return total

    @staticmethod
    def _join_side() -> None:
        """The weight-gradient side stream (rave_amd.ops._OnSide) is joined by a callback at the end of every backward pass;
        joining again before the optimizer reads the gradients is a no-op then, and the safety net when a pass ended
        without running its callbacks."""
        from . import ops
        ops.join_side_streams()

    def split_features(self, features):
        """rave/model.py:276-286."""
        feature_real, feature_fake = [], []
        for scale in features:
            true, fake = zip(*map(lambda x: torch.split(x, x.shape[0] // 2, 0), scale))
            feature_real.append(true)
            feature_fake.append(fake)
        return feature_real, feature_fake

    # ---- rave/model.py:288-424
    def training_step(self, batch, batch_idx, eps: Optional[torch.Tensor] = None, grad_sync=None,
                      capture_safe: bool = False, grad_begin=None):
        """``eps`` injects the reparametrisation noise (parity runs); ``grad_begin(optimizer_index)`` is called right
        after that optimizer's ``zero_grad()`` and ``grad_sync(optimizer_index)`` between backward and optimizer.step
        (data-parallel gradient averaging: rave_amd.ddp.GradReducer.begin / finish).
        ``capture_safe``: no host synchronisation inside the step (the reference's ``reg.item()`` test,
        rave/model.py:393, becomes "always add the regulariser" -- identical whenever it is non-zero, i.e. always
        for the variational encoder), so that the step can be recorded into a hipGraph."""
        gen_opt, dis_opt = self.optimizers()
        # all weight-normalised convs: weight norm + MFMA repack refreshed in two launches (plumbing;
        # the reference's weight_norm pre-hooks do the same work layer by layer)
        if batch.is_cuda:
            # a fresh zeroed pool of range slots for this step's activations (ops.range_reset: one fill launch; recorded into
            # a hipGraph it re-zeroes them at every replay)
            from . import ops as _ops
            _ops.range_reset(batch.device)
            # the discriminator only runs (and only then normalises its weights) in phase 2
            self.prepare_weights(with_discriminator=bool(self.warmed_up))
        x_raw = batch
        x_raw.requires_grad = True
        batch_size = x_raw.shape[:-2]
        self.encoder.set_warmed_up(self.warmed_up)
        self.decoder.set_warmed_up(self.warmed_up)

        x_multiband = _pqmf_encode(self.pqmf, x_raw)                   # (= self.encode(x_raw, return_mb=True), unrolled)
        rf = (0, 0)
        if self.valid_signal_crop:                                     # rave/model.py:321-329
            # the buffer is read on the host (as the reference does); while a hipGraph is being recorded the value
            # seen by the preceding eager iterations is used (it only changes in validation)
            if not (batch.is_cuda and torch.cuda.is_current_stream_capturing()):
                self._rf_host = tuple(int(v) for v in self.receptive_field.tolist())
            rf = getattr(self, "_rf_host", (0, 0))
        x_mb_loss = valid_signal_crop(x_multiband, rf[0], rf[1]) if rf[0] + rf[1] else x_multiband
        z = self.encoder(x_multiband)
        z, reg = self.encoder.reparametrize(z, eps)[:2]

        y = self.decoder(z)
        y_multiband = y[..., :x_multiband.shape[-1]]
        if rf[0] + rf[1]:
            y_multiband = valid_signal_crop(y_multiband, rf[0], rf[1])
        x_multiband = x_mb_loss

        # the generator loss terms as (value, first factor): on the GPU the products and the weighted sum of
        # rave/model.py:336-344, 392-412 are ONE launch each way (ops.loss_combine, same bits as the ATen chain below)
        terms = {}
        # Round 5: the multiband distance runs BESIDE the PQMF synthesis and the fullband distance, on the stream the
        # weight-gradient branch uses later in the pass (idle until then; still two streams in the recorded graph).  autograd
        # runs a node's backward on the stream its forward ran on, so the two distances overlap in both directions: 9.74-9.75 ms
        # per step against 9.95-10.01 in three alternated pairs (profiles/round5_ab_knobs_and_negative_results.txt); same bits
        # (no atomics anywhere).  RH_LOSS_SIDE_STREAM=0 / RH_BWD_SIDE_STREAM=0: one stream.
        import os
        loss_side = None
        if batch.is_cuda and os.environ.get("RH_LOSS_SIDE_STREAM", "1") != "0":
            from . import ops
            loss_side = ops._side_stream(batch.device) if ops._side_enabled() else None
        if loss_side is not None:
            loss_side.wait_stream(torch.cuda.current_stream(batch.device))
            with torch.cuda.stream(loss_side):
                multiband_distance = self.multiband_audio_distance(x_multiband, y_multiband)
            for t_ in (x_multiband, y_multiband):
                t_.record_stream(loss_side)
            # the distances were allocated from the side stream's pool and are consumed on the compute stream (loss_combine)
            for v_ in multiband_distance.values():
                if torch.is_tensor(v_):
                    v_.record_stream(torch.cuda.current_stream(batch.device))
        else:
            multiband_distance = self.multiband_audio_distance(x_multiband, y_multiband)
        for k, v in multiband_distance.items():
            terms[f"multiband_{k}"] = (v, self.weights["multiband_audio_distance"])
        y_raw = _pqmf_decode(self.pqmf, y, batch_size=batch_size, n_channels=self.n_channels)
        y_raw = y_raw[..., :x_raw.shape[-1]]
        fullband_distance = self.audio_distance(x_raw, y_raw)
        if loss_side is not None:
            torch.cuda.current_stream(batch.device).wait_stream(loss_side)
        for k, v in fullband_distance.items():
            terms[f"fullband_{k}"] = (v, self.weights["audio_distance"])

        feature_matching_distance = 0.
        dis_step = bool(self.warmed_up) and not (batch_idx % self.update_discriminator_every)
        frozen = []
        if self.warmed_up:
            if self.skip_dead_grads and dis_step:
                xy = torch.cat([x_raw.detach(), y_raw.detach()], 0)
            else:
                xy = torch.cat([x_raw, y_raw], 0)
            if self.skip_dead_grads and not dis_step:
                frozen = [q for q in self.discriminator.parameters() if q.requires_grad]
            try:
                for q in frozen:
                    q.requires_grad_(False)
                features = self.discriminator(xy)
            finally:
                for q in frozen:
                    q.requires_grad_(True)
            feature_real, feature_fake = self.split_features(features)
            loss_dis = 0
            loss_adv = 0
            fused_fm = self._fused_feature_matching(features)
            for scale_real, scale_fake in zip(feature_real, feature_fake):
                if fused_fm is None:
                    current = sum(map(self.feature_matching_fun,
                                      scale_real[self.num_skipped_features:],
                                      scale_fake[self.num_skipped_features:])) / len(scale_real[self.num_skipped_features:])
                    feature_matching_distance = feature_matching_distance + current
                _dis, _adv = self.gan_loss(scale_real[-1], scale_fake[-1])
                loss_dis = loss_dis + _dis
                loss_adv = loss_adv + _adv
            feature_matching_distance = fused_fm if fused_fm is not None else feature_matching_distance / len(feature_real)
        else:
            loss_dis = torch.zeros((), device=x_raw.device, dtype=x_raw.dtype)     # (no host-to-device copy)
            loss_adv = torch.zeros((), device=x_raw.device, dtype=x_raw.dtype)

        if capture_safe:
            # per-step host scalars must not be baked into a recorded graph: beta_factor is changed every step by the
            # reference's BetaWarmupCallback (rave/model.py:83-107), so the captured step reads it from a 0-d device
            # tensor that GraphedTrainingStep refreshes (fill_) before every replay -- same f32 product as `reg * float`
            terms["regularization"] = (reg, self.beta_device(reg.device))
        elif reg.item():
            terms["regularization"] = (reg, self.beta_factor)
        if self.warmed_up:
            terms["feature_matching"] = (feature_matching_distance, self.weights["feature_matching"])
            terms["adversarial"] = (loss_adv, self.weights["adversarial"])
        fused_sum = _loss_combine_fused() and all(torch.is_tensor(v) and v.is_cuda and v.dtype == torch.float32
                                                  and v.numel() == 1 for v, _ in terms.values()) and 0 < len(terms) <= 16
        loss_gen = {}
        loss_gen_value = None
        if fused_sum:
            from . import ops
            names = list(terms)
            w1 = [0. if torch.is_tensor(terms[k][1]) else terms[k][1] for k in names]
            w1_dev = [terms[k][1] if torch.is_tensor(terms[k][1]) else None for k in names]
            loss_gen_value, scaled = ops.loss_combine([terms[k][0] for k in names], w1, [self.weights.get(k, 1.) for k in names],
                                                      w1_dev)
            for i, k in enumerate(names):
                loss_gen[k] = scaled[i]
        else:
            for k, (v, w) in terms.items():
                loss_gen[k] = (v * w) if torch.is_tensor(w) else (w * v)

        if dis_step:
            dis_opt.zero_grad()
            if grad_begin is not None:
                grad_begin(1)
            loss_dis.backward()
            if grad_sync is not None:
                grad_sync(1)
            self._join_side()
            dis_opt.step()
        else:
            gen_opt.zero_grad()
            if grad_begin is not None:
                grad_begin(0)
            if loss_gen_value is None:
                loss_gen_value = 0.
                for k, v in loss_gen.items():
                    loss_gen_value += v * self.weights.get(k, 1.)
            loss_gen_value.backward()
            if grad_sync is not None:
                grad_sync(0)
            self._join_side()
            gen_opt.step()

        self.release_weights()
        if self._prep is not None:       # (fused optimizers do not bump parameter version counters)
            for pr in self._prep:
                pr.invalidate()
        self.logged = dict(loss_gen)
        self.logged["loss_dis"] = loss_dis
        return self.logged


    # ---- rave/model.py:426-443
    def validation_step(self, x, batch_idx, eps: Optional[torch.Tensor] = None):
        """Returns (cat([x, y], -1), latent mean or None), as the reference; the validation distance is kept in
        ``self.logged["validation"]`` (the reference logs it through the trainer).  ``eps`` injects the reparametrisation
        draw (parity runs).  Lightning calls this under ``model.eval()``: AdaptiveInstanceNormalization sites then run
        their statistics / transfer branch (blocks.AdaptiveInstanceNormalization)."""
        if x.is_cuda:
            self.prepare_weights(reuse=True)
        z = self.encode(x)
        if isinstance(self.encoder, blocks.VariationalEncoder):
            mean = torch.split(z, z.shape[1] // 2, 1)[0]
        else:
            mean = None
        z = self.encoder.reparametrize(z, eps)[0]
        y = self.decode(z)
        distance = self.audio_distance(x, y)
        # a NEW dict (the current one may be the object a GraphedTrainingStep hands back at every replay: writing into it would
        # leak a stale entry into all later training logs -- ADVICE r5), the value detached
        self.logged = dict(self.logged, validation=sum(distance.values()).detach())
        self.release_weights()
        return torch.cat([x, y], -1), mean


class LinearLR:
    """torch.optim.lr_scheduler.LinearLR(start_factor, end_factor, total_iters) in closed form:
    lr_t = base * (start + (end - start) * min(t, total) / total).  A learning rate that lives on the device
    (capturable optimizers) is updated IN PLACE, so that a recorded hipGraph keeps reading the current value
    (torch's scheduler rebinds ``param_group["lr"]`` to a new object)."""

    def __init__(self, optimizer, start_factor: float, end_factor: float, total_iters: int):
        self.opt, self.start, self.end, self.total = optimizer, start_factor, end_factor, max(int(total_iters), 1)
        self.base = [float(g["lr"]) for g in optimizer.param_groups]
        self.t = 0

    def get_last_lr(self):
        f = self.start + (self.end - self.start) * min(self.t, self.total) / self.total
        return [b * f for b in self.base]

    def step(self):
        self.t += 1
        for g, lr in zip(self.opt.param_groups, self.get_last_lr()):
            if torch.is_tensor(g["lr"]):
                g["lr"].fill_(lr)
            else:
                g["lr"] = lr


class GraphedTrainingStep:
    """The static training step as ONE hipGraph launch: prepare_weights -> forward -> losses
    -> backward -> Adam of ``RAVE.training_step`` recorded once per (phase, step kind) for a fixed batch shape through
    ``torch.cuda.CUDAGraph`` (hipGraph on ROCm) and replayed.  ~650 kernel launches per step leave the host: the eager
    step is within a few % of being launch-bound on the host (9.3 ms of pure CPU time per step measured).

    Eager ``training_step`` stays the parity path; tests assert bit-identical parameters after 8 steps of both.
    The model must have been set up with ``configure_optimizers(capturable=True)``.

    What a recorded step reads at REPLAY time (device memory, updated in place): parameters, optimizer state, the
    learning rates (LinearLR), ``beta_factor`` (RAVE.beta_device, refreshed here before every replay) and the input
    batch / noise.  What is BAKED IN at capture time: the loss ``weights``, the phase (``warmed_up``) and step kind
    (one graph per (phase, kind) key), ``valid_signal_crop``'s receptive field, every buffer-derived host flag.
    Data-dependent one-time initialisation cannot be recorded: a model whose RVQ codebooks are not initialised yet
    (``EuclideanCodebook.inited == 0`` with the quantizer enabled) is refused -- run eager steps until the k-means
    init has happened, then construct / call this class."""

    def __init__(self, model: "RAVE", example_batch: torch.Tensor, inject_eps: bool = False, warmup_iters: int = 3,
                 grad_begin=None, grad_sync=None, before_step=None):
        """``grad_begin`` / ``grad_sync``: the data-parallel reducer's begin / finish (rave_amd.ddp.GradReducer), recorded
        INTO the graph -- the bucket all-reduces are stream-ordered RCCL launches, capturable like any kernel, so a
        data-parallel rank replays one graph per step exactly like a single-GPU run.  ``before_step``: called at the
        start of the recorded region (rave_amd.ddp.BufferSync.sync: the per-step buffer broadcast)."""
        self.model = model
        self.sync_kw = dict(grad_begin=grad_begin, grad_sync=grad_sync)
        self.before_step = before_step
        self.x = example_batch.detach().clone()
        self.eps = None
        if inject_eps:
            self.eps = torch.zeros(self.x.shape[0], model.latent_size, self.x.shape[-1] // self._hop(), device=self.x.device)
        self.warmup_iters = warmup_iters
        self.graphs = {}
        self.logged = {}

    def _hop(self) -> int:
        # one dry encode to learn the latent rate; train-mode side effects (v1 BatchNorm running statistics, lazily
        # initialised buffers) are undone so that the snapshot taken before the capture is the caller's state
        saved = {k: v.clone() for k, v in self.model.named_buffers()}
        with torch.no_grad():
            z = self.model.encode(self.x[:1])
            for k, v in self.model.named_buffers():
                v.copy_(saved[k])
        _reset_host_shadows(self.model)
        return self.x.shape[-1] // z.shape[-1]

    def _check_capturable(self) -> None:
        m = self.model
        enc = getattr(m, "encoder", None)
        enabled = getattr(enc, "enabled", None)
        if enabled is not None and bool(enabled):
            for name, mod in m.named_modules():
                inited = getattr(mod, "inited", None)
                if torch.is_tensor(inited) and not bool(inited):
                    raise RuntimeError(
                        f"GraphedTrainingStep: {name}.inited == 0 -- the RVQ k-means initialisation is host-driven, "
                        "data-dependent work that a recorded graph cannot contain; run eager training_step()s until the "
                        "codebooks are initialised, then capture")

    def _key(self, batch_idx: int):
        m = self.model
        return bool(m.warmed_up) and not (batch_idx % m.update_discriminator_every)

    def capture(self, batch: torch.Tensor, batch_idx: int, eps: Optional[torch.Tensor] = None) -> None:
        """Records the graph of this (phase, step kind) WITHOUT replaying it: the warm-up iterations run (and are rolled back),
        the capture executes nothing.  Data-parallel callers use it to agree across ranks that every rank recorded its step
        before any rank replays one -- a rank that replays while another fell back to the eager step would wait in a
        collective nobody else enters."""
        self(batch, batch_idx, eps, _replay=False)

    def __call__(self, batch: torch.Tensor, batch_idx: int, eps: Optional[torch.Tensor] = None, _replay: bool = True):
        m = self.model
        key = (bool(m.warmed_up), self._key(batch_idx))
        with torch.no_grad():            # (training_step marks its input as requiring grad, rave/model.py:292)
            self.x.copy_(batch)
            if self.eps is not None:
                self.eps.copy_(eps)
        if key not in self.graphs:
            self._check_capturable()
            # a few eager iterations on a side stream first (allocator warm-up, lazy one-time initialisations),
            # restoring parameters / optimizer state afterwards so that the capture does not change the trajectory
            state = ({k: v.clone() for k, v in m.state_dict().items()},
                     [_clone_opt(o) for o in m.optimizers()])
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(self.warmup_iters):
                    if self.before_step is not None:
                        self.before_step()
                    m.training_step(self.x, batch_idx, eps=self.eps, capture_safe=True, **self.sync_kw)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            m.load_state_dict(state[0])
            for o, st in zip(m.optimizers(), state[1]):
                _restore_opt(o, st)
            for o in m.optimizers():
                o.zero_grad(set_to_none=True)
            # host shadows of flag buffers were filled by the warm-up run; the restore may have changed the buffers
            _reset_host_shadows(m)
            self._check_capturable()
            for mod in m.modules():          # host-side caches keyed on parameter versions (the restore bumped them)
                if hasattr(mod, "refresh_host_caches"):
                    mod.refresh_host_caches()
            m.set_phase_flags_eagerly()
            g = torch.cuda.CUDAGraph()
            import os
            kw = {}
            if torch.distributed.is_available() and torch.distributed.is_initialized():
                # The process group's watchdog thread wakes every 100 ms and queries the end events of the collectives it
                # still lists -- those of the eager warm-up steps, long complete.  This HIP runtime refuses hipEventQuery on an
                # event whose STREAM has meanwhile joined a capture (hipErrorCapturedEvent, although the event was recorded
                # before), which the watchdog turns into an abort of the process: about one data-parallel capture in eight
                # (profiles/round6_rccl_watchdog_abort.txt).  Let it retire the finished collectives before the recording starts.
                import time
                torch.cuda.synchronize()
                time.sleep(0.5)
            if os.environ.get("RH_GRAPH_PRIORITY", "0") == "1":
                # experiment: record the step on a HIGH-priority stream, so that the data-gradient chain (the critical path)
                # wins the arbitration against the weight-gradient branch, which forks onto a default-priority stream
                kw["stream"] = torch.cuda.Stream(priority=-1)
            if torch.distributed.is_available() and torch.distributed.is_initialized():
                # RCCL's watchdog thread polls the events of earlier collectives (hipEventQuery) whenever it likes; under the
                # default capture_error_mode = "global" such a call from ANOTHER thread while this one records is an error, which
                # the watchdog turns into an abort of the process (ProcessGroupNCCL.cpp: Watchdog::run) -- intermittently, about
                # one data-parallel capture in four on these boxes (round 5).  "thread_local" restricts the check to this thread.
                kw["capture_error_mode"] = "thread_local"
            # A step that cannot be recorded (e.g. a collective of a backend without capture support) must leave the process fit for
            # the caller's eager fallback: (1) the streams the refused step forked are joined back INSIDE the recording -- ending a
            # capture with unjoined work fails and, on this runtime, leaves every stream of it in capture mode for good
            # (tools/debug/gloo_capture_recover.py); (2) the refused call's error code is read away from the runtime's per-thread
            # "last error", where the next checked call of this thread would trip over it.
            err = None
            try:
                with torch.cuda.graph(g, **kw):
                    try:
                        if self.before_step is not None:
                            self.before_step()
                        logged = m.training_step(self.x, batch_idx, eps=self.eps, capture_safe=True, **self.sync_kw)
                    except Exception as e:        # noqa: BLE001 -- handed on below, after the recording has been closed
                        err = e
                        from . import ops as _ops
                        _ops.abandon_side_streams(rejoin=True)
            except Exception as e:                # noqa: BLE001 -- the recording could not be closed either
                err = err or e
            if err is not None:
                from . import ops as _ops
                _ops.abandon_side_streams()
                for _ in range(3):
                    try:
                        torch.cuda.synchronize()
                        torch.zeros(1, device=self.x.device)
                        break
                    except Exception:             # noqa: BLE001 -- the stale error itself
                        continue
                raise err
            # the capture itself does not execute anything: parameters are still the restored ones
            self.graphs[key] = (g, logged)
        g, logged = self.graphs[key]
        if not _replay:
            return logged
        m.beta_device(self.x.device)     # per-step scalar the recorded step reads from device memory
        g.replay()
        if m._prep is not None:          # the replayed optimizer changed parameters behind the version counters
            for pr in m._prep:
                pr.invalidate()
        self.logged = logged
        return logged


def _reset_host_shadows(model: nn.Module) -> None:
    """Forget every host-side shadow of a flag buffer (blocks.host_flag / blocks._set_warmed_up): the next read goes to
    the device again.  Needed after anything that rewrites buffers behind the modules' back (load_state_dict)."""
    for mod in model.modules():
        mod.__dict__.pop("_host_flags", None)
        if "_warmed_up_host" in mod.__dict__:
            mod._warmed_up_host = None


def _clone_opt(opt):
    return {id(p): {k: (v.clone() if torch.is_tensor(v) else v) for k, v in st.items()} for p, st in opt.state.items()}


def _restore_opt(opt, saved):
    # two passes: state tensors may be SHARED between parameters (an optimizer may alias one step counter into several
    # state[p]["step"]) -- first reset whatever the warm-up iterations created, then restore the saved values, so that a
    # restored counter is never zeroed afterwards through an alias (dict order must not decide)
    for p in list(opt.state.keys()):
        if id(p) not in saved:
            # state created by the warm-up iterations: keep the tensors (a state created INSIDE the capture would be
            # re-initialised by every replay) but reset them to the freshly-initialised values (all zero)
            for k, v in opt.state[p].items():
                if torch.is_tensor(v):
                    v.zero_()
    for p in list(opt.state.keys()):
        if id(p) in saved:
            for k, v in saved[id(p)].items():
                if torch.is_tensor(v):
                    opt.state[p][k].copy_(v)
                else:
                    opt.state[p][k] = v


V2_DILATIONS = [[1, 3, 9], [1, 3, 9], [1, 3, 9], [1, 3]]


def build_v1(n_channels: int = 1, capacity: int = 64, ratios=(4, 4, 4, 2), latent_size: int = 128,
             n_band: int = 16, sampling_rate: int = 44100, use_noise: bool = True) -> "RAVE":
    """configs/v1.gin: v1 Encoder (BatchNorm) / Generator (ResidualStack, loudness + noise branches),
    multi-scale discriminator only, feature matching L1 x10, discriminator every 2nd step."""
    cc.set_default_padding_mode("centered")
    blocks.set_normalization_mode("weight_norm")
    ratios = list(ratios)
    enc = partial(blocks.VariationalEncoder,
                  encoder=partial(blocks.Encoder, data_size=n_band, capacity=capacity, latent_size=latent_size,
                                  ratios=ratios, n_out=2, sample_norm=False, repeat_layers=1))
    dec = partial(blocks.Generator, latent_size=latent_size, capacity=capacity, data_size=n_band, ratios=ratios,
                  loud_stride=1, use_noise=use_noise)
    msd = partial(discriminator.MultiScaleDiscriminator, n_discriminators=3,
                  convnet=partial(discriminator.ConvNet, out_size=1, capacity=capacity, n_layers=4, stride=4,
                                  conv=nn.Conv1d, kernel_size=15))
    stft = partial(losses.MultiScaleSTFT, scales=[2048, 1024, 512, 256, 128], sample_rate=sampling_rate, magnitude=True)
    dist = partial(losses.AudioDistanceV1, multiscale_stft=stft, log_epsilon=1e-7)
    return RAVE(latent_size=latent_size, sampling_rate=sampling_rate,
                pqmf=partial(pqmf.CachedPQMF, attenuation=100, n_band=n_band), encoder=enc, decoder=dec,
                discriminator=msd, phase_1_duration=1000000, gan_loss=losses.hinge_gan, valid_signal_crop=False,
                feature_matching_fun=partial(losses.mean_difference, norm="L1"), num_skipped_features=0,
                audio_distance=dist, multiband_audio_distance=dist, weights={"feature_matching": 10},
                update_discriminator_every=2, n_channels=n_channels, n_bands=n_band)


def build_v2_small(**kw) -> "RAVE":
    """configs/v2_small.gin: CAPACITY 48, RATIOS [4,2,2,2], NoiseGeneratorV2, discriminator every 2nd step."""
    d = dict(capacity=48, ratios=(4, 2, 2, 2), noise=True, update_discriminator_every=2)
    d.update(kw)
    return build_v2(**d)


def build_v2(n_channels: int = 1, capacity: int = 96, ratios=(4, 4, 4, 2), latent_size: int = 128,
             n_band: int = 16, dilations=None, sampling_rate: int = 44100, causal: bool = False,
             disc_capacity: Optional[int] = None, snake: bool = False, adain: bool = False,
             noise: bool = False, update_discriminator_every: int = 4, discriminator_kind: str = "v2",
             encoder_kind: str = "variational", noise_augmentation: int = 0, num_quantizers: int = 16,
             codebook_size: int = 1024, log_epsilon: float = 1e-7, num_skipped_features: int = 1,
             spectral_capacity: int = 32) -> RAVE:
    """configs/v1.gin + configs/v2.gin transcribed (cf. oracle/ref_models.py for the citations).
    ``snake`` / ``adain`` add the generator-side overlays of configs/v3.gin (snake.gin: every activation
    -> blocks.Snake; adain.gin: AdaptiveInstanceNormalization before every unit); ``causal`` =
    configs/causal.gin.  ``discriminator_kind``: "v2" (MPD + MSD, v2.gin:53-75), "descript"
    (descript_discriminator.gin), "spectral" (spectral_discriminator.gin: MSD + Encodec STFT nets).
    ``encoder_kind``: "variational" or "discrete" (discrete.gin: EncoderV2(n_out=1) + RVQ + noise channels)."""
    cc.set_default_padding_mode("causal" if causal else "centered")
    blocks.set_normalization_mode("weight_norm")
    dil = dilations or V2_DILATIONS
    ratios = list(ratios)
    extra = {}
    blocks.set_dilated_unit_activation(blocks.Snake if snake else None)
    if snake:
        extra["activation"] = blocks.Snake
    if adain:
        extra["adain"] = blocks.AdaptiveInstanceNormalization
    if encoder_kind == "variational":
        enc = partial(blocks.VariationalEncoder,
                      encoder=partial(blocks.EncoderV2, data_size=n_band, capacity=capacity, ratios=ratios,
                                      latent_size=latent_size, n_out=2, kernel_size=3, dilations=dil, **extra))
    elif encoder_kind == "discrete":   # configs/discrete.gin:24-40
        enc = partial(blocks.DiscreteEncoder,
                      encoder_cls=partial(blocks.EncoderV2, data_size=n_band, capacity=capacity, ratios=ratios,
                                          latent_size=latent_size, n_out=1, kernel_size=3, dilations=dil, **extra),
                      vq_cls=partial(quantization.ResidualVectorQuantization, num_quantizers=num_quantizers,
                                     dim=latent_size, codebook_size=codebook_size),
                      num_quantizers=num_quantizers, noise_augmentation=noise_augmentation)
    else:
        raise ValueError(encoder_kind)
    dec_latent = latent_size + noise_augmentation      # core.get_augmented_latent_size (v2.gin:24-26,47)
    if noise:   # configs/v2_small.gin:42-57
        extra_dec = dict(noise_module=partial(blocks.NoiseGeneratorV2, hidden_size=64, data_size=n_band,
                                              ratios=[2, 2, 2], noise_bands=32))
    else:
        extra_dec = {}
    dec = partial(blocks.GeneratorV2, data_size=n_band, capacity=capacity, ratios=ratios,
                  latent_size=dec_latent, kernel_size=3, dilations=dil, amplitude_modulation=True, **extra,
                  **extra_dec)
    common = dict(out_size=1, capacity=disc_capacity or capacity, n_layers=4, stride=4)
    mpd = partial(discriminator.MultiPeriodDiscriminator, periods=[2, 3, 5, 7, 11],
                  convnet=partial(discriminator.ConvNet, conv=nn.Conv2d, kernel_size=(5, 1), **common))
    msd = partial(discriminator.MultiScaleDiscriminator, n_discriminators=3,
                  convnet=partial(discriminator.ConvNet, conv=nn.Conv1d, kernel_size=15, **common))
    if discriminator_kind == "v2":
        disc = partial(discriminator.CombineDiscriminators, discriminators=[mpd, msd])
    elif discriminator_kind == "descript":
        disc = descript_discriminator.DescriptDiscriminator
    elif discriminator_kind == "spectral":   # configs/spectral_discriminator.gin
        mssd = partial(discriminator.MultiScaleSpectralDiscriminator, scales=[4096, 2048, 1024, 512, 256],
                       convnet=partial(discriminator.EncodecConvNet, capacity=spectral_capacity))
        disc = partial(discriminator.CombineDiscriminators, discriminators=[msd, mssd])
    else:
        raise ValueError(discriminator_kind)
    stft = partial(losses.MultiScaleSTFT, scales=[2048, 1024, 512, 256, 128], sample_rate=sampling_rate,
                   magnitude=True)
    dist = partial(losses.AudioDistanceV1, multiscale_stft=stft, log_epsilon=log_epsilon)
    model = RAVE(latent_size=latent_size, sampling_rate=sampling_rate,
                 pqmf=partial(pqmf.CachedPQMF, attenuation=100, n_band=n_band),
                 encoder=enc, decoder=dec, discriminator=disc, phase_1_duration=1000000,
                 gan_loss=losses.hinge_gan, valid_signal_crop=True,
                 feature_matching_fun=partial(losses.mean_difference, norm="L1", relative=True),
                 num_skipped_features=num_skipped_features, audio_distance=dist, multiband_audio_distance=dist,
                 weights={"feature_matching": 20}, update_discriminator_every=update_discriminator_every,
                 n_channels=n_channels, n_bands=n_band)
    cc.set_default_padding_mode("centered")
    blocks.set_dilated_unit_activation(None)
    return model


def build_v3(n_channels: int = 2, causal: bool = True, **kw) -> RAVE:
    """configs/v3.gin (+ causal.gin; BASELINE config 5 is stereo): v2 + adain.gin + snake.gin +
    descript_discriminator.gin."""
    d = dict(n_channels=n_channels, causal=causal, snake=True, adain=True, discriminator_kind="descript")
    d.update(kw)
    return build_v2(**d)


def build_discrete(spectral: bool = True, **kw) -> RAVE:
    """configs/discrete.gin (+ spectral_discriminator.gin; BASELINE config 4): RATIOS [4,4,2,2], EncoderV2(n_out=1)
    + 16-stage RVQ (1024 codes) + 128 noise channels, log_epsilon 1, num_skipped_features 0."""
    d = dict(ratios=(4, 4, 2, 2), encoder_kind="discrete", noise_augmentation=128, log_epsilon=1.0,
             num_skipped_features=0, discriminator_kind="spectral" if spectral else "v2")
    d.update(kw)
    return build_v2(**d)
