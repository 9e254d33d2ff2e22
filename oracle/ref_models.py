"""TEST INFRASTRUCTURE ONLY.

Builds the reference's own ``rave.RAVE`` (through oracle/ref_import.py) with the
bindings of the shipped .gin files transcribed as explicit ``functools.partial``
keyword arguments (the gin shim is Tier A, i.e. no parser).
Every block cites the .gin lines it transcribes.  ``capacity`` & friends can be
shrunk to produce the small golden fixtures of tests/golden.
"""
from functools import partial

import torch
import torch.nn as nn

from ref_import import import_reference, set_causal

V2_DILATIONS = [[1, 3, 9], [1, 3, 9], [1, 3, 9], [1, 3]]  # configs/v2.gin:13-18


def build_reference_rave(config="v2", n_channels=1, capacity=None, ratios=None,
                         latent_size=128, n_band=16, causal=False, dilations=None,
                         sampling_rate=44100):
    """Reference ``rave.RAVE`` for config in {'v2', 'v2_small', 'v3'}."""
    rave = import_reference()
    from rave import blocks, core, discriminator, pqmf
    set_causal(causal)

    if config == "v1":
        # configs/v1.gin
        capacity = capacity or 64
        ratios = ratios or [4, 4, 4, 2]
        import gin
        enc = partial(blocks.VariationalEncoder,
                      encoder=partial(blocks.Encoder, data_size=n_band, capacity=capacity, latent_size=latent_size,
                                      ratios=ratios, sample_norm=False, repeat_layers=1, n_out=2))      # v1.gin:44-55
        gin.bind("ResidualStack", kernel_sizes=[3], dilations_list=[[1, 1], [3, 1], [5, 1]])             # v1.gin:67-69
        gin.bind("NoiseGenerator", ratios=[4, 4, 4], noise_bands=5)                                       # v1.gin:71-73
        dec = partial(blocks.Generator, latent_size=latent_size, capacity=capacity, data_size=n_band,
                      ratios=ratios, loud_stride=1, use_noise=True)                                       # v1.gin:58-65
        msd = partial(discriminator.MultiScaleDiscriminator, n_discriminators=3,
                      convnet=partial(discriminator.ConvNet, out_size=1, capacity=capacity, n_layers=4, stride=4,
                                      conv=nn.Conv1d, kernel_size=15))                                    # v1.gin:75-88
        stft = partial(core.MultiScaleSTFT, scales=[2048, 1024, 512, 256, 128], sample_rate=sampling_rate, magnitude=True)
        dist = partial(core.AudioDistanceV1, multiscale_stft=stft, log_epsilon=1e-7)
        return rave.RAVE(latent_size=latent_size, sampling_rate=sampling_rate,
                         pqmf=partial(pqmf.CachedPQMF, attenuation=100, n_band=n_band), encoder=enc, decoder=dec,
                         discriminator=msd, phase_1_duration=1000000, gan_loss=core.hinge_gan,
                         valid_signal_crop=False, feature_matching_fun=partial(core.mean_difference, norm="L1"),
                         num_skipped_features=0, audio_distance=dist, multiband_audio_distance=dist,
                         weights={"feature_matching": 10}, update_discriminator_every=2,
                         n_channels=n_channels, n_bands=n_band)
    if config in ("v2", "v3"):
        capacity = capacity or 96                       # v2.gin:20
        ratios = ratios or [4, 4, 4, 2]                 # v2.gin:19
        dil = dilations or V2_DILATIONS
        update_every = 4                                # v2.gin:83
    elif config == "v2_small":
        capacity = capacity or 48                       # v2_small.gin:20
        ratios = ratios or [4, 2, 2, 2]                 # v2_small.gin:19
        dil = dilations or [[1, 3, 9], [1, 3, 9], [1, 3, 9], [1, 3]]
        update_every = 2                                # v2_small.gin:93
    else:
        raise ValueError(config)

    act_kwargs = {}
    adain = None
    unit_defaults = blocks.DilatedUnit.__init__.__defaults__
    if config == "v3":
        # configs/snake.gin: every activation -> blocks.Snake ; configs/adain.gin.  gin binds
        # blocks.DilatedUnit.activation separately (EncoderV2/GeneratorV2 do not forward theirs):
        # emulate that binding by swapping the default of DilatedUnit.__init__ during construction.
        act_kwargs = dict(activation=blocks.Snake)
        adain = blocks.AdaptiveInstanceNormalization
        blocks.DilatedUnit.__init__.__defaults__ = (blocks.Snake,)

    enc_kwargs = dict(data_size=n_band, capacity=capacity, ratios=ratios,
                      latent_size=latent_size, n_out=2, kernel_size=3,
                      dilations=dil, **act_kwargs)           # v2.gin:30-37
    dec_kwargs = dict(data_size=n_band, capacity=capacity, ratios=ratios,
                      latent_size=latent_size, kernel_size=3, dilations=dil,
                      amplitude_modulation=True, **act_kwargs)  # v2.gin:43-50
    if adain is not None:
        enc_kwargs["adain"] = adain
        dec_kwargs["adain"] = adain
    if config == "v2_small":
        # v2_small.gin:42-57
        dec_kwargs["noise_module"] = partial(blocks.NoiseGeneratorV2, hidden_size=64,
                                             data_size=n_band, ratios=[2, 2, 2],
                                             noise_bands=32)

    encoder = partial(blocks.VariationalEncoder,
                      encoder=partial(blocks.EncoderV2, **enc_kwargs))  # v2.gin:39-40
    decoder = partial(blocks.GeneratorV2, **dec_kwargs)

    convnet_common = dict(out_size=1, capacity=capacity, n_layers=4, stride=4)  # v1.gin:75-80
    if config == "v3":
        # configs/descript_discriminator.gin
        from rave import descript_discriminator as dd
        disc = partial(dd.DescriptDiscriminator, rates=[], periods=[2, 3, 5, 7, 11],
                       fft_sizes=[2048, 1024, 512], sample_rate=sampling_rate,
                       bands=[(0.0, 0.1), (0.1, 0.25), (0.25, 0.5), (0.5, 0.75), (0.75, 1.0)])
    else:
        mpd = partial(discriminator.MultiPeriodDiscriminator, periods=[2, 3, 5, 7, 11],
                      convnet=partial(discriminator.ConvNet, conv=nn.Conv2d,
                                      kernel_size=(5, 1), **convnet_common))  # v2.gin:53-64
        msd = partial(discriminator.MultiScaleDiscriminator, n_discriminators=3,
                      convnet=partial(discriminator.ConvNet, conv=nn.Conv1d,
                                      kernel_size=15, **convnet_common))      # v1.gin:82-88
        disc = partial(discriminator.CombineDiscriminators, discriminators=[mpd, msd])  # v2.gin:70-75

    stft = partial(core.MultiScaleSTFT, scales=[2048, 1024, 512, 256, 128],
                   sample_rate=sampling_rate, magnitude=True)                  # v1.gin:25-28
    dist = partial(core.AudioDistanceV1, multiscale_stft=stft, log_epsilon=1e-7)  # v1.gin:21-23, v2.gin:23

    model = rave.RAVE(
        latent_size=latent_size, sampling_rate=sampling_rate,
        pqmf=partial(pqmf.CachedPQMF, attenuation=100, n_band=n_band),       # v1.gin:37-39
        encoder=encoder, decoder=decoder, discriminator=disc,
        phase_1_duration=1000000,                                            # v1.gin:18
        gan_loss=core.hinge_gan, valid_signal_crop=True,                     # v1.gin:101, v2.gin:81
        feature_matching_fun=partial(core.mean_difference, norm="L1", relative=True),  # v1.gin:90-91, v2.gin:77-78
        num_skipped_features=1,                                              # v2.gin:82
        audio_distance=dist, multiband_audio_distance=dist,
        weights={"feature_matching": 20},                                    # v2.gin:85-87
        update_discriminator_every=update_every,
        n_channels=n_channels, n_bands=n_band,
    )
    blocks.DilatedUnit.__init__.__defaults__ = unit_defaults
    return model


def attach_optimizers(model):
    """What Lightning does with ``configure_optimizers`` (rave/model.py:226-236)."""
    gen, dis = model.configure_optimizers()
    model._opts = (gen["optimizer"], dis["optimizer"])
    model._scheds = gen["lr_scheduler"]["scheduler"]
    return model


def build_reference_discrete(capacity=96, latent_size=128, noise_augmentation=128, num_quantizers=16, codebook_size=1024,
                             spectral_capacity=32, n_channels=1, n_band=16, sampling_rate=44100):
    """Reference ``rave.RAVE`` for configs/discrete.gin + configs/spectral_discriminator.gin (BASELINE configs[3]):
    RATIOS [4,4,2,2] (discrete.gin:14), EncoderV2(n_out=1) inside DiscreteEncoder with a ResidualVectorQuantization
    bottleneck and noise channels (discrete.gin:24-40), generator latent = latent + noise (v2.gin:24-26,47),
    log_epsilon = 1 (discrete.gin:22), num_skipped_features = 0 (discrete.gin:48), discriminators =
    [MultiScaleDiscriminator, MultiScaleSpectralDiscriminator] (spectral_discriminator.gin:6-17)."""
    rave = import_reference()
    from rave import blocks, core, discriminator, pqmf, quantization
    set_causal(False)
    ratios = [4, 4, 2, 2]
    enc = partial(blocks.DiscreteEncoder,
                  encoder_cls=partial(blocks.EncoderV2, data_size=n_band, capacity=capacity, ratios=ratios,
                                      latent_size=latent_size, n_out=1, kernel_size=3, dilations=V2_DILATIONS),
                  vq_cls=partial(quantization.ResidualVectorQuantization, num_quantizers=num_quantizers, dim=latent_size,
                                 codebook_size=codebook_size),
                  num_quantizers=num_quantizers, noise_augmentation=noise_augmentation)
    dec = partial(blocks.GeneratorV2, data_size=n_band, capacity=capacity, ratios=ratios,
                  latent_size=latent_size + noise_augmentation, kernel_size=3, dilations=V2_DILATIONS,
                  amplitude_modulation=True)
    msd = partial(discriminator.MultiScaleDiscriminator, n_discriminators=3,
                  convnet=partial(discriminator.ConvNet, out_size=1, capacity=capacity, n_layers=4, stride=4,
                                  conv=nn.Conv1d, kernel_size=15))
    mssd = partial(discriminator.MultiScaleSpectralDiscriminator, scales=[4096, 2048, 1024, 512, 256],
                   convnet=partial(discriminator.EncodecConvNet, capacity=spectral_capacity))
    disc = partial(discriminator.CombineDiscriminators, discriminators=[msd, mssd])
    stft = partial(core.MultiScaleSTFT, scales=[2048, 1024, 512, 256, 128], sample_rate=sampling_rate, magnitude=True)
    dist = partial(core.AudioDistanceV1, multiscale_stft=stft, log_epsilon=1)
    return rave.RAVE(latent_size=latent_size, sampling_rate=sampling_rate,
                     pqmf=partial(pqmf.CachedPQMF, attenuation=100, n_band=n_band), encoder=enc, decoder=dec,
                     discriminator=disc, phase_1_duration=200000, gan_loss=core.hinge_gan, valid_signal_crop=True,
                     feature_matching_fun=partial(core.mean_difference, norm="L1", relative=True),
                     num_skipped_features=0, audio_distance=dist, multiband_audio_distance=dist,
                     weights={"feature_matching": 20}, update_discriminator_every=4, n_channels=n_channels,
                     n_bands=n_band)
