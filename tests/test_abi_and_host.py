"""CPU tests: the C-ABI library loads and exports every symbol declared in include/rave_hip.h (no
compute calls without a GPU), descriptor-only entry points behave, and the host-side mirrors keep
the reference's interface (padding rule, state_dict layout, error behaviour)."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "rave_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rh_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from rave_amd import _lib as L
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L.lib, n), f"librave_hip.so does not export {n}"
    assert L.lib.rh_version() >= 100


def test_descriptor_queries_and_errors():
    from rave_amd import _lib as L
    d = L.ConvDesc(batch=32, c_in=96, c_out=96, l_in=4096, l_out=4096, kernel=3, stride=1, dilation=9,
                   pad_left=9, transposed=0, groups=1, inner=1, in_valid=0, act=1, act_slope=0.2)
    # K in blocks of 16: f32 operand + the x6 section (P pieces x 2 bytes per weight: P = 2 f16 pieces in the product build,
    # 3 bf16 pieces in the comparison build) + the 16-byte range record behind the fragments
    P = 2 if L.lib.rh_x6_uses_ranges() else 3
    n = 3 * 96 * 96
    assert L.lib.rh_conv1d_packed_floats(C.byref(d), 0) == n + n * P // 2 + 4
    assert L.lib.rh_conv1d_packed_floats(C.byref(d), 1) == n + n * P // 2 + 4
    d0 = L.ConvDesc(batch=32, c_in=96, c_out=192, l_in=4096, l_out=1024, kernel=8, stride=4, dilation=1,
                    pad_left=3, transposed=0, groups=1, inner=1, in_valid=0, act=1, act_slope=0.2)
    # strided: phase-interleaved octets (forward); the data gradient carries two bf16x6 sections of the same size
    # (per-phase taps and the "virtual rows" form with contiguous output runs, chosen per launch)
    n = 8 * 96 * 192
    assert L.lib.rh_conv1d_packed_floats(C.byref(d0), 0) == n + n * P // 2 + 4
    assert L.lib.rh_conv1d_packed_floats(C.byref(d0), 1) == n + 2 * (n * P // 2) + 4
    # transposed forward: same two sections (per output phase + virtual rows); its data gradient is a strided gather
    up = L.ConvDesc(batch=32, c_in=192, c_out=96, l_in=1024, l_out=4096, kernel=8, stride=4, dilation=1,
                    pad_left=2, transposed=1, groups=1, inner=1, in_valid=0, act=1, act_slope=0.2)
    assert L.lib.rh_conv1d_packed_floats(C.byref(up), 0) == n + 2 * (n * P // 2) + 4
    assert L.lib.rh_conv1d_packed_floats(C.byref(up), 1) == n + n * P // 2 + 4
    # k = 5, stride 4 (period discriminators): phases carry 2 / 1 / 1 / 1 taps, the virtual-row section pads them to 2
    mpd = L.ConvDesc(batch=128, c_in=96, c_out=192, l_in=8192, l_out=2048, kernel=5, stride=4, dilation=1,
                     pad_left=2, transposed=0, groups=1, inner=1, in_valid=0, act=1, act_slope=0.2)
    n32 = 5 * 192 * 96
    assert L.lib.rh_conv1d_packed_floats(C.byref(mpd), 1) == n32 + n32 * P // 2 + (192 // 16) * 2 * (2 * P) * (96 * 4) * 4 + 4
    d1 = L.ConvDesc(batch=2, c_in=1, c_out=96, l_in=4096, l_out=1024, kernel=15, stride=4, dilation=1,
                    pad_left=7, transposed=0, groups=1, inner=1, in_valid=0, act=0, act_slope=0.0)
    assert L.lib.rh_conv1d_packed_floats(C.byref(d1), 0) == 15 * 1 * 96       # C*stride % 16 != 0: f32 operand only
    assert L.lib.rh_conv1d_workspace_bytes(C.byref(d)) > 0
    assert L.lib.rh_conv1d_fwd_workspace_bytes(C.byref(d)) == 0            # large grid: no split-K
    d2 = L.ConvDesc(batch=32, c_in=1536, c_out=256, l_in=32, l_out=32, kernel=3, stride=1, dilation=1,
                    pad_left=1, transposed=0, groups=1, inner=1, in_valid=0, act=1, act_slope=0.2)
    assert L.lib.rh_conv1d_fwd_workspace_bytes(C.byref(d2)) > 0            # short sequence: split-K scratch
    bad = L.ConvDesc(batch=1, c_in=4, c_out=4, l_in=8, l_out=8, kernel=3, stride=1, dilation=1, pad_left=1,
                     transposed=0, groups=2, inner=1, in_valid=0, act=0, act_slope=0.0)
    assert L.lib.rh_conv1d_packed_floats(C.byref(bad), 0) == -1            # groups != 1 -> unsupported
    rc = L.lib.rh_conv1d_pack_f32(C.byref(bad), None, None, None, None)
    assert rc == -2 and b"groups" in L.lib.rh_last_error()
    with pytest.raises(RuntimeError):
        L.check(rc, "pack")


def test_kernel_family_queries():
    """Which kernel a launch takes is a pure function of the descriptor (no GPU needed): bf16x6 for the generator
    convs and their strided / transposed relatives, the vector-ALU kernels for 1- / 2-channel first layers, f32 MFMA
    for what neither takes; the environment switches flip them."""
    from rave_amd import _lib as L
    fam = lambda d, which: L.lib.rh_conv1d_kernel_family(C.byref(d), which, 0, 0)
    unit = L.ConvDesc(batch=32, c_in=96, c_out=96, l_in=4096, l_out=4096, kernel=3, stride=1, dilation=9,
                      pad_left=9, transposed=0, groups=1, inner=1, in_valid=0, act=1, act_slope=0.2)
    down = L.ConvDesc(batch=32, c_in=96, c_out=192, l_in=4096, l_out=1024, kernel=8, stride=4, dilation=1,
                      pad_left=3, transposed=0, groups=1, inner=1, in_valid=0, act=1, act_slope=0.2)
    first = L.ConvDesc(batch=64, c_in=1, c_out=96, l_in=65536, l_out=16384, kernel=15, stride=4, dilation=1,
                       pad_left=7, transposed=0, groups=1, inner=1, in_valid=0, act=0, act_slope=0.0)
    out = L.ConvDesc(batch=32, c_in=96, c_out=32, l_in=4096, l_out=4096, kernel=7, stride=1, dilation=1,
                     pad_left=3, transposed=0, groups=1, inner=1, in_valid=0, act=1, act_slope=0.2)
    assert [fam(unit, 0), fam(unit, 1), L.lib.rh_conv1d_bwd_weight_kernel_family(C.byref(unit))] == [1, 1, 1]
    # <= 64 gradient rows on long sequences: the f32 MFMA weight-gradient kernel
    assert [fam(out, 0), fam(out, 1), L.lib.rh_conv1d_bwd_weight_kernel_family(C.byref(out))] == [1, 1, 0]
    assert [fam(down, 0), fam(down, 1), L.lib.rh_conv1d_bwd_weight_kernel_family(C.byref(down))] == [1, 1, 1]
    assert [fam(first, 0), fam(first, 1), L.lib.rh_conv1d_bwd_weight_kernel_family(C.byref(first))] == [2, 2, 2]
    assert L.lib.rh_conv1d_workspace_bytes(C.byref(first)) > 0          # partial tiles of the fused weight + bias gradient
    os.environ["RH_SMALLC"] = "0"
    try:
        assert 2 not in [fam(first, 0), fam(first, 1), L.lib.rh_conv1d_bwd_weight_kernel_family(C.byref(first))]
    finally:
        del os.environ["RH_SMALLC"]
    os.environ["RH_CONV_X6"] = "0"
    try:
        assert [fam(unit, 0), fam(down, 1)] == [0, 0]
    finally:
        del os.environ["RH_CONV_X6"]


def test_mfma_kernels_use_no_scratch():
    """The bf16x6 / LDS-DMA conv and weight-gradient kernels must keep their accumulators in registers: the code
    objects' metadata (private segment size, spill counts) is read back from the built library."""
    import importlib.util
    from pathlib import Path
    spec = importlib.util.spec_from_file_location(
        "kernel_resources", Path(__file__).resolve().parents[1] / "tools" / "kernel_resources.py")
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    ks = kr.kernels()
    assert any("conv_x6_kernel" in k["name"] for k in ks) and any("wgrad_x6_kernel" in k["name"] for k in ks)
    bad = kr.scratch_kernels()
    assert not bad, bad
    for k in ks:
        if "conv_x6_kernel" in k["name"]:
            assert k["vgpr"] + k["agpr"] <= 256, k      # two workgroups of four waves per CU


def test_get_padding_matches_cached_conv_rule():
    from rave_amd import cc
    import rave_oracle as O
    for k in (1, 2, 3, 4, 7, 8, 15, 33, 513):
        for d in (1, 3, 9):
            for mode in ("centered", "causal"):
                assert cc.get_padding(k, dilation=d, mode=mode) == O.get_padding(k, dilation=d, mode=mode)
    assert cc.get_padding(513) == (256, 256) and cc.get_padding(33) == (16, 16)
    cc.set_default_padding_mode("causal")
    try:
        assert cc.get_padding(513) == (512, 0)
        assert cc.get_padding(15, 4, mode="centered") == (7, 7)          # explicit mode wins (discriminator.py:91-97)
    finally:
        cc.set_default_padding_mode("centered")


def test_module_tree_and_state_dict_match_reference_layout(golden_dir):
    """Keys and shapes of the drop-in modules == the reference's state_dict (golden fixture)."""
    from rave_amd import model as M
    g = torch.load(os.path.join(golden_dir, "v2_tiny.pt"), weights_only=False)
    c = g["config"]
    m = M.build_v2(capacity=c["capacity"], latent_size=c["latent_size"])
    sd = m.state_dict()
    for k, v in g["state_dict"].items():
        assert k in sd, k
        assert tuple(sd[k].shape) == tuple(v.shape), k
    extra = set(sd) - set(g["state_dict"])
    assert extra <= {"receptive_field"}
    missing, unexpected = m.load_state_dict(g["state_dict"], strict=False)
    assert not unexpected
    # attributes scripts/export_onnx.py:34-60 reads
    conv = m.encoder.encoder.net[0]
    assert conv._pad == (3, 3) and conv.cumulative_delay == 0 and conv.bias is None
    assert conv.weight_g.shape == (c["capacity"], 1, 1)
    up = m.decoder.net[2]
    assert up.weight_g.shape[0] == up.in_channels          # ConvTranspose: norm over dim 0 = IN channels
    assert not m.pqmf.forward_conv.weight.requires_grad


def test_no_cpu_fallback():
    from rave_amd import ops
    g = ops.ConvGeom(pad_left=1, pad_right=1)
    with pytest.raises(RuntimeError, match="GPU"):
        ops.conv1d(torch.zeros(1, 4, 8), torch.zeros(4, 4, 3), None, geom=g)
    with pytest.raises(RuntimeError):
        ops.pqmf_analysis(torch.zeros(1, 1, 64), torch.zeros(16, 1, 513), (256, 256))


def test_streaming_mode_is_refused():
    from rave_amd import cc
    cc.use_cached_conv(False)
    with pytest.raises(NotImplementedError):
        cc.use_cached_conv(True)


def _have_reference():
    import ref_import
    return ref_import.reference_available()


@pytest.mark.skipif(not _have_reference(), reason="/root/reference not present (GPU box)")
def test_latent_wrappers_match_reference():
    """WasserteinEncoder / SphericalEncoder / DiscreteEncoder(disabled) reparametrize == reference
    (rave/blocks.py:748-849); pure latent-space torch maths, no HIP call."""
    from ref_import import import_reference
    import_reference()
    from rave import blocks as rb
    from rave_amd import blocks as hb
    enc = lambda n_channels=1: torch.nn.Identity()
    z = torch.randn(3, 8, 16)
    eps = torch.randn(3 * 16, 8)
    noise = torch.randn(3, 4, 16)
    w_ref, w_hip = rb.WasserteinEncoder(enc, noise_augmentation=0), hb.WasserteinEncoder(enc, noise_augmentation=0)
    torch.manual_seed(1)
    zr, rr = w_ref.reparametrize(z)
    torch.manual_seed(1)
    zh, rh = w_hip.reparametrize(z)
    assert torch.equal(zr, zh) and torch.allclose(rr, rh)
    s_ref, s_hip = rb.SphericalEncoder(enc), hb.SphericalEncoder(enc)
    assert torch.equal(s_ref.reparametrize(z)[0], s_hip.reparametrize(z)[0])
    d_ref = rb.DiscreteEncoder(enc, vq_cls=lambda: torch.nn.Identity(), num_quantizers=16, noise_augmentation=4)
    d_hip = hb.DiscreteEncoder(enc, vq_cls=None, num_quantizers=16, noise_augmentation=4)
    torch.manual_seed(2)
    a, da = d_ref.reparametrize(z)
    torch.manual_seed(2)
    b, db = d_hip.reparametrize(z)
    assert torch.equal(a, b) and float(da) == float(db) == 0.0
    assert set(d_ref.state_dict()) == set(d_hip.state_dict())


def test_conv2d_discriminator_mirrors_have_reference_state_dict_layout(golden_dir):
    """Module tree / parameter names / shapes of the spectral and descript discriminator mirrors equal the
    reference's (recorded in the golden fixture by oracle/make_golden.py); no GPU compute."""
    import os
    from functools import partial
    import torch
    from rave_amd import descript_discriminator as DD, discriminator as D
    g = torch.load(os.path.join(golden_dir, "disc2d_tiny.pt"), weights_only=False)
    c = g["encodec"]["config"]
    m = D.MultiScaleSpectralDiscriminator(c["scales"], partial(D.EncodecConvNet, capacity=c["capacity"]),
                                          n_channels=c["n_channels"])
    assert {k: tuple(v.shape) for k, v in m.state_dict().items() if not k.endswith(".window")} == g["encodec"]["shapes"]
    assert sum(k.endswith(".window") for k in m.state_dict()) == len(c["scales"])
    c = g["descript"]["config"]
    m = DD.DescriptDiscriminator(periods=c["periods"], fft_sizes=c["fft_sizes"], n_channels=c["n_channels"])
    assert {k: tuple(v.shape) for k, v in m.state_dict().items() if not k.endswith(".window")} == g["descript"]["shapes"]
    with __import__("pytest").raises(RuntimeError):
        m(torch.zeros(1, 2, 2048))      # CPU tensors: the HIP path has no CPU fallback


_INTEGRATION_SCRIPT = r'''
import json, sys
sys.dont_write_bytecode = True
root, use_hip = sys.argv[1], sys.argv[2] == "hip"
sys.path.insert(0, root); sys.path.insert(0, root + "/oracle")
if use_hip:                                  # INTEGRATION.md section 1: swap the operator package before `import rave`
    import rave_amd.cc as cc_hip
    sys.modules["cached_conv"] = cc_hip
from ref_import import import_reference
rave = import_reference()
import torch
from rave import blocks, pqmf
torch.manual_seed(0)
dil = [[1, 3, 9], [1, 3, 9], [1, 3, 9], [1, 3]]
mods = dict(
    enc=blocks.VariationalEncoder(lambda n_channels=1: blocks.EncoderV2(data_size=16, capacity=8, ratios=[4, 4, 4, 2],
        latent_size=16, n_out=2, kernel_size=3, dilations=dil, n_channels=n_channels)),
    dec=blocks.GeneratorV2(data_size=16, capacity=8, ratios=[4, 4, 4, 2], latent_size=16, kernel_size=3, dilations=dil,
        amplitude_modulation=True),
    pq=pqmf.CachedPQMF(attenuation=100, n_band=16))
out = {k: {n: list(t.shape) for n, t in m.state_dict().items()} for k, m in mods.items()}
import cached_conv
convs = [type(m).__module__ for m in mods["enc"].modules() if type(m).__name__ in ("Conv1d", "ConvTranspose1d")]
print("RESULT" + json.dumps(dict(shapes=out, conv_modules=sorted(set(convs)), cc=cached_conv.__name__)))
'''


def test_reference_blocks_build_on_the_drop_in_operator_package():
    """INTEGRATION.md section 1, exercised: with ``sys.modules['cached_conv'] = rave_amd.cc`` the UNMODIFIED reference
    ``rave.blocks`` / ``rave.pqmf`` construct on the HIP operator classes and expose exactly the state_dict layout
    they have on the (restated) cached_conv package.  Construction only -- no compute without a GPU."""
    import json, os, subprocess, sys
    import pytest
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.isdir("/root/reference/rave"):
        pytest.skip("reference tree not present (GPU box)")

    def run(kind):
        r = subprocess.run([sys.executable, "-c", _INTEGRATION_SCRIPT, root, kind], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT")][-1][6:])

    ref, hip = run("shim"), run("hip")
    assert hip["cc"] == "rave_amd.cc" and ref["cc"] == "cached_conv"
    assert hip["conv_modules"] == ["rave_amd.cc"]
    # buffers the real package may add for streaming ("...pad") are not part of either build
    assert hip["shapes"] == ref["shapes"]


def test_valid_signal_crop_and_lr_schedule_match_reference():
    """rave/core.py:220-225 and rave/model.py:234-236,272-274 restated in rave_amd/model.py."""
    import torch
    from rave_amd import model as M
    x = torch.arange(2 * 16 * 100, dtype=torch.float32).reshape(2, 16, 100)
    lf, rf = torch.tensor(1234), torch.tensor(567)
    got = M.valid_signal_crop(x, lf, rf)
    assert torch.equal(got, x[..., 1234 // 16:-567 // 16])
    assert torch.equal(M.valid_signal_crop(x, torch.tensor(40), torch.tensor(0)), x[..., 2:])
    from ref_import import reference_available, import_reference
    if reference_available():
        rave = import_reference()
        assert torch.equal(got, rave.core.valid_signal_crop(x, lf, rf))
    # generator LR: LinearLR 1.0 -> 0.1 over phase_1_duration steps, stepped once per batch
    lin = torch.nn.Linear(2, 2)

    class Tiny(M.RAVE):
        def __init__(self):
            torch.nn.Module.__init__(self)
            self.encoder, self.decoder, self.discriminator = lin, torch.nn.Linear(2, 2), torch.nn.Linear(2, 2)
            self.warmup = 10
            self._opts = self._gen_sched = None

    m = Tiny()
    gen_opt, dis_opt = m.optimizers()
    for _ in range(5):
        gen_opt.step()
        m.on_train_batch_end(None, None, 0)
    assert abs(gen_opt.param_groups[0]["lr"] - 1e-3 * (1.0 - 0.9 * 5 / 10)) < 1e-12
    assert dis_opt.param_groups[0]["lr"] == 1e-4
    for _ in range(10):
        gen_opt.step()
        m.on_train_batch_end(None, None, 0)
    assert abs(gen_opt.param_groups[0]["lr"] - 1e-4) < 1e-12


_GIN_SCRIPT = r'''
import json, os, sys
root, overlay = sys.argv[1], sys.argv[2] == "1"
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle"))
sys.dont_write_bytecode = True
from ref_import import import_reference
rave = import_reference()
import gin, torch
gin.clear_config()
files = ["configs/v2.gin"] + ([os.path.join(root, "rave_amd", "configs", "mi355x.gin")] if overlay else [])
gin.parse_config_files_and_bindings(files, ["CAPACITY = 8", "LATENT_SIZE = 16"])
torch.manual_seed(0)
model = rave.RAVE()                       # the reference's own class, every argument from the parsed .gin files
mods = {k: type(getattr(model, k)).__module__ for k in ("pqmf", "encoder", "decoder", "discriminator")}
inner = sorted({type(m).__module__.split(".")[0] for m in model.modules()
                if type(m).__name__ in ("Conv1d", "ConvTranspose1d", "Conv2dK1", "PlainConv1d", "Conv2d")})
sd = {k: list(v.shape) for k, v in model.state_dict().items()}
print("RESULT" + json.dumps(dict(mods=mods, conv_roots=inner, sd=sd, skipped=gin.SKIPPED, cls=type(model).__module__,
                                 crop=bool(model.valid_signal_crop), every=model.update_discriminator_every)))
'''


def test_gin_overlay_builds_the_unmodified_rave_on_the_drop_in_modules():
    """rave_amd/configs/mi355x.gin, parsed AFTER the reference's own configs/v2.gin (which includes v1.gin) by the gin
    shim, makes the UNMODIFIED ``rave.model.RAVE`` construct its pqmf / encoder / decoder / discriminator from the
    HIP drop-in classes -- with exactly the state_dict keys and shapes the stock configuration produces.
    (Construction only: no GPU here, and /root/reference does not exist on the GPU box.)"""
    import json, os, subprocess, sys
    import pytest
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.isdir("/root/reference/rave"):
        pytest.skip("reference tree not present (GPU box)")

    def run(overlay):
        r = subprocess.run([sys.executable, "-c", _GIN_SCRIPT, root, "1" if overlay else "0"], capture_output=True, text=True,
                           timeout=600, env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
        assert r.returncode == 0, r.stderr[-3000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT")][-1][6:])

    ref, hip = run(False), run(True)
    assert ref["cls"] == hip["cls"] == "rave.model"                    # the reference's own RAVE in both builds
    assert set(ref["mods"].values()) <= {"rave.pqmf", "rave.blocks", "rave.discriminator"}
    assert hip["mods"] == {"pqmf": "rave_amd.pqmf", "encoder": "rave_amd.blocks", "decoder": "rave_amd.blocks",
                           "discriminator": "rave_amd.discriminator"}
    assert hip["conv_roots"] == ["rave_amd"]                           # no torch / cached_conv convolution left
    assert hip["crop"] and hip["every"] == 4                           # v2.gin's own rave.RAVE bindings still apply
    drop = lambda sd: {k: v for k, v in sd.items() if not k.endswith(("paddings.0.pad", ".pad"))}
    assert drop(hip["sd"]) == drop(ref["sd"])
    assert all(s.startswith("dataset.") for s in hip["skipped"])       # only rave.dataset (udls / lmdb) is out of reach


def test_dense_batch_major_layout_check():
    """Host logic of the fused feature-matching op (rave_amd.ops._dense_batch_major): which feature-map layouts can be used
    in their own memory order -- dim 0 outermost over one dense block, any order of the other dims."""
    import torch
    from rave_amd.ops import _dense_batch_major as ok
    b, p, c, h = 8, 7, 12, 5
    base = torch.zeros(b * p, c, h)
    assert ok(base.view(b, p * c * h)) and ok(base.view(b, p, c, h))
    view = base.view(b, p, c, h).permute(0, 2, 3, 1)            # what the period discriminators hand out
    assert not view.is_contiguous() and ok(view)
    assert torch.empty_like(view).stride() == view.stride()     # the gradient buffer takes the same memory order
    assert not ok(base.view(b, p, c, h).permute(1, 0, 2, 3))    # batch not outermost
    assert not ok(base.view(b, p, c, h)[:, :, :, :3])           # not dense
    assert not ok(base.view(b, p, c, h)[:, :3])                 # gaps between batch items
    assert not ok(torch.zeros(7, 4))                            # odd batch: no real / fake halves
    assert ok(torch.zeros(2, 1, 1, 9).permute(0, 3, 1, 2))      # size-1 dims do not matter


def test_stft_loss_launch_plans_cover_every_geometry():
    """Host side of the in-kernel STFT distance (stft_loss.hip) over many (n_fft, length, rows): the forward workgroups cover
    every frame; the backward workgroups tile the padded row in whole hop blocks, the first one owns the left margin and
    what it folds onto (> n_fft samples), the last one the right margin (>= 6 blocks), and the LDS image fits."""
    import ctypes as C
    from rave_amd import _lib as L
    out = (C.c_int64 * 6)()
    checked = 0
    for n_fft in (128, 256, 512, 1024, 2048):
        hop = n_fft // 4
        for t_len in (n_fft // 2 + 1, n_fft, 1100, 4096, 5000, 65536, 65536 + 4, 131072 + hop - 1, 1 << 20):
            for rows in (1, 2, 32, 512, 4096):
                if not L.lib.rh_stft_loss_supported(n_fft, hop, t_len, rows):
                    assert t_len <= n_fft // 2
                    continue
                assert L.lib.rh_stft_loss_plan_info(n_fft, t_len, rows, out) == 0
                fpw, fwd_wgs, cb, nch, last, lds = list(out)
                nf = t_len // hop + 1
                n_blocks = (t_len + n_fft + hop - 1) // hop
                g = 256 // (n_fft // 8)
                assert fpw % g == 0 and fwd_wgs * fpw >= nf > (fwd_wgs - 1) * fpw
                assert (nch - 1) * cb + last == n_blocks and last >= 6 and (nch == 1 or cb >= 6)
                assert min(cb, n_blocks) * hop > n_fft                 # first workgroup: left margin + its fold targets
                assert lds <= 160 * 1024
                assert L.lib.rh_stft_loss_workspace_bytes(n_fft, t_len, rows) == rows * fwd_wgs * 12
                checked += 1
    assert checked > 150
    assert L.lib.rh_stft_loss_plan_info(2048, 1024, 4, out) < 0       # reflect padding needs t > n_fft / 2


def test_conv2d_x6_tile_plans_cover_every_geometry():
    """Host side of the bf16x6 general Conv2d (conv2d_x6.hip) through rh_conv2d_plan_info, forward and data gradient of the
    real Encodec / MRD layer geometries at several plane sizes and batches plus awkward ones: the tile grid covers the
    per-phase output grid, the patch image fits the LDS and the conversion-task slots, every tap offset stays inside the
    patch, channel counts outside blocks of 16 stay on the f32 kernels, and the packed operand grows by the bf16x6 section."""
    import ctypes as C
    from rave_amd import _lib as L
    out = (C.c_int64 * 16)()
    layers = [((9, 3), (2, 1), (1, 1), (4, 1)), ((9, 3), (2, 1), (1, 2), (4, 2)), ((9, 3), (2, 1), (1, 4), (4, 4)),
              ((3, 3), (1, 1), (1, 1), (1, 1)), ((3, 9), (1, 2), (1, 1), (1, 4)), ((3, 9), (1, 1), (1, 1), (1, 4)),
              ((5, 1), (3, 1), (1, 1), (2, 0)), ((4, 3), (3, 2), (2, 2), (3, 2)), ((2, 2), (4, 1), (1, 1), (0, 0)),
              ((1, 1), (1, 1), (1, 1), (0, 0))]
    planes = [(2049, 61), (1025, 125), (513, 253), (129, 1021), (33, 509), (17, 1021), (129, 26), (129, 102), (7, 6), (9, 1)]
    checked = x6 = smallm = 0
    for (kh, kw), (sh, sw), (dh, dw), (ph, pw) in layers:
        for h, w in planes:
            ho = (h + 2 * ph - dh * (kh - 1) - 1) // sh + 1
            wo = (w + 2 * pw - dw * (kw - 1) - 1) // sw + 1
            if h + 2 * ph - dh * (kh - 1) - 1 < 0 or w + 2 * pw - dw * (kw - 1) - 1 < 0:
                continue
            for batch, ci, co in ((64, 32, 32), (2, 32, 1), (3, 48, 96), (1, 16, 40), (2, 2, 32), (2, 24, 32)):
                d = L.Conv2dDesc(batch=batch, c_in=ci, c_out=co, h_in=h, w_in=w, h_out=ho, w_out=wo, kh=kh, kw=kw, sh=sh, sw=sw,
                                 dh=dh, dw=dw, ph=ph, pw=pw, act=1, act_slope=0.2)
                for which in (0, 1):
                    assert L.lib.rh_conv2d_plan_info(C.byref(d), which, out) == 0
                    fam, tm, tn, nq, TR, TQ, nb, lds, wgs, PH, PW, P, maxoff, nphase, tiles_r, tiles_q = list(out)
                    cin = ci if which == 0 else co
                    m = co if which == 0 else ci
                    base = kh * kw * cin * ((m + 31) // 32 * 32)
                    packed = L.lib.rh_conv2d_packed_floats(C.byref(d), which)
                    checked += 1
                    valu_shape = (sh, sw) == (1, 1) and dh == 1 and kh in (3, 9)
                    few_in = which == 0 and cin <= 4 and 4 < m <= 32        # first conv of a stack, forward (conv2d_smallc.hip)
                    if fam == 2:          # <= 4 output rows (conv2d_smallm.hip) or <= 4 input channels: vector-ALU kernels
                        assert valu_shape and (m <= 4 or few_in)
                        smallm += 1
                        continue
                    assert not (valu_shape and (m <= 4 or few_in))
                    if cin % 16:
                        assert fam == 0 and packed == base
                        continue
                    # + the matrix-core fragments (two f16 pieces = 4 bytes per weight; the comparison build: three bf16 pieces = 6)
                    # + the 16-byte range record behind them
                    frag = base if L.lib.rh_x6_uses_ranges() == 1 else base // 2 * 3
                    assert packed == base + frag + 4
                    if fam == 0:          # tensors beyond the 2 GiB buffer descriptors, or no tile fits
                        assert 4 * batch * max(ci * h * w, co * ho * wo) >= 2 ** 31 - 1 or P == 0
                        continue
                    x6 += 1
                    rows, qcols = (ho, wo) if which == 0 else (-(-h // sh), -(-w // sw))
                    assert tm in (1, 2, 3) and tn in (1, 2) and nq in (4, 8) and 32 * tm * (-(-m // (32 * tm))) >= m
                    assert TR * TQ * nb == 128 * tn and TQ <= 32
                    assert tiles_r * TR >= rows and tiles_q * TQ >= qcols and (tiles_r - 1) * TR < rows and (tiles_q - 1) * TQ < qcols
                    assert 2 * P <= 256 * nq and P == nb * PH * PW
                    pieces = 2 if L.lib.rh_x6_uses_ranges() == 1 else 3          # 16-byte fragments per octet (common.hpp: kX6P)
                    assert lds == (2 * 2 * pieces * 32 * tm + 2 * pieces * P) * 16 and lds <= 160 * 1024
                    is_h, is_w = (sh, sw) if which == 0 else (1, 1)
                    # the farthest fragment a lane reads: last row / column of the tile + the largest tap offset
                    assert (nb - 1) * PH * PW + (TR - 1) * is_h * PW + (TQ - 1) * is_w + maxoff < P
                    assert nphase == (1 if which == 0 else sh * sw)
                    assert wgs == -(-batch // nb) * tiles_r * tiles_q * (-(-m // (32 * tm))) * nphase
    assert checked > 500 and x6 > 200 and smallm > 20


def test_table_entry_points_validate_their_host_tables_without_a_gpu():
    """The by-value-table entry points of the backward tail (collected weight-norm backward, loss arithmetic, one finalize for
    all STFT scales, Adam with per-parameter step counters): an empty table is a no-op, a malformed one is refused with a
    message BEFORE anything is launched -- host logic only, no GPU needed."""
    from rave_amd import _lib as L
    assert L.lib.rh_weight_norm_bwd_batched_f32(None, 0, None) == 0
    it = (L.WnBwdItem * 1)()
    it[0].rows, it[0].cols = 4, 3                        # null pointers
    assert L.lib.rh_weight_norm_bwd_batched_f32(it, 1, None) != 0 and b"bad item" in L.lib.rh_last_error()
    assert L.lib.rh_weight_norm_bwd_batched_f32(None, 2, None) != 0
    li = (L.LossItem * 17)()
    buf = (C.c_float * 32)()
    addr = C.addressof(buf)
    assert L.lib.rh_loss_combine_fwd_f32(li, 0, addr, addr, None) != 0 and b"terms" in L.lib.rh_last_error()
    assert L.lib.rh_loss_combine_fwd_f32(li, 17, addr, addr, None) != 0              # more than 16 terms
    assert L.lib.rh_loss_combine_fwd_f32(li, 2, addr, addr, None) != 0 and b"null value" in L.lib.rh_last_error()
    assert L.lib.rh_loss_combine_bwd_f32(li, 2, addr, addr, None) != 0
    pp = (C.c_void_p * 9)(*([addr] * 9))
    nb = (C.c_int64 * 9)(*([12] * 9))
    assert L.lib.rh_stft_loss_finalize_all_f32(pp, nb, 9, addr, addr, addr, None) != 0 and b"scales" in L.lib.rh_last_error()
    nb[0] = 10                                                                        # not a multiple of 3 floats
    assert L.lib.rh_stft_loss_finalize_all_f32(pp, nb, 2, addr, addr, addr, None) != 0 and b"partials" in L.lib.rh_last_error()
    assert L.lib.rh_adam_step_f32(None, 0, addr, 0.5, 0.9, 1e-8, addr, None) == 0    # no items: nothing launched
    ai = (L.AdamItem * 1)()
    ai[0].p = ai[0].g = ai[0].m = ai[0].v = addr
    ai[0].n = 8                                                                       # no step counter
    assert L.lib.rh_adam_step_f32(ai, 1, addr, 0.5, 0.9, 1e-8, addr, None) != 0 and b"bad item" in L.lib.rh_last_error()
    assert L.lib.rh_conv1d_bwd_weight_wn_fused_launches() == 0


def test_product_pqmf_filter_bank_is_bit_identical_to_the_reference_buffers():
    """SURVEY section 8 rows a2-a4 (rave/pqmf.py:55-89,245-254: kaiser_filter -> get_prototype -> get_qmf_bank ->
    center_pad_next_pow_2, polyphase weights): the buffers ``rave_amd.pqmf.CachedPQMF(100, 16)`` builds ON ITS OWN -- every
    GPU parity test overwrites them with oracle / golden weights, so a scipy upgrade could silently change the product's
    filters (VERDICT r5 weak #1c) -- against the reference's buffers in tests/golden/pqmf.pt (oracle/make_golden.py)."""
    from rave_amd.pqmf import CachedPQMF
    g = torch.load(os.path.join(os.path.dirname(__file__), "golden", "pqmf.pt"))
    m = CachedPQMF(100, 16)
    assert torch.equal(m.h, g["h"]) and torch.equal(m.hk, g["hk"])
    assert torch.equal(m.forward_conv.weight.detach(), g["w_fwd"])
    assert torch.equal(m.inverse_conv.weight.detach(), g["w_inv"])
