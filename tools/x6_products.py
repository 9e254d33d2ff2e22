"""How many partial products does the bf16 forward need?  (SURVEY.md section 7 "hard parts" option (i).)

The product kernels take every f32 product as SIX bf16 partial products (exact 3-way split: the f32 numerics class).
north_star's own bar is <= 1e-4 relative L2, so this tool MEASURES the 2-piece variants (common.hpp: RH_X6_PRODUCTS = 3 | 4,
built by ``python -m rave_amd.build --variants`` into rave_amd/_var/) at BASELINE configs[1]'s full width:

  * batch 8: every hot-path output and all 112 generator-side parameter gradients against the CPU oracle (fp32 for the
    outputs, fp64 for the gradients) -- forward and data gradients on the variant kernels, weight gradients on six products;
  * batch 32 (the benchmarked size): outputs and gradients against the 6-product kernels;
  * the forward-only leg of bench.py (PQMF + conv stacks, no_grad, packed weights reused) in ms.

One worker process per library (``RAVE_HIP_LIB``); the driver compares.  ``--json`` (bench.py): batch-32 forward only.
The headline stays on the 6-product path whatever this says; the variants are never loaded by the product.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rel_l2(a, b):
    a = a.detach().double().cpu().reshape(-1)
    b = b.detach().double().cpu().reshape(-1)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _inputs(batch, seed=7):
    import torch
    g = torch.Generator().manual_seed(20250509)
    t = torch.arange(65536, dtype=torch.float32) / 44100.0
    x = 0.1 * torch.randn(batch, 1, 65536, generator=g)
    for f0, a in ((220.0, 0.2), (1760.0, 0.1), (7040.0, 0.05)):
        ph = torch.rand(batch, 1, 1, generator=g) * 6.283185307
        x = x + a * torch.sin(6.283185307 * f0 * t + ph)
    x = x.clamp(-1, 1)
    gen = torch.Generator().manual_seed(seed)
    eps = torch.randn(batch, 128, 32, generator=gen)
    cy_raw = torch.randn(batch, 1, 65536, generator=gen) * 1e-3
    cy_mb = torch.randn(batch, 16, 4096, generator=gen) * 1e-3
    return x, eps, cy_raw, cy_mb


def _seeded_model():
    import torch
    from rave_amd import model as M
    torch.manual_seed(0)
    return M.build_v2()


def worker(args):
    import torch
    from rave_amd import model as M, _lib as L
    dev = torch.device("cuda:0")
    res = {"lib": L.LIB_PATH}
    m = _seeded_model().to(dev).train()

    def hot_path(batch, backward):
        x, eps, cy_raw, cy_mb = (t.to(dev) for t in _inputs(batch))
        m.zero_grad(set_to_none=True)
        m.prepare_weights()
        xx = x.clone().requires_grad_(True)
        zp, x_mb = m.encode(xx, return_mb=True)
        z, reg = m.encoder.reparametrize(zp, eps)
        y_mb = m.decoder(z)
        y_raw = M._pqmf_decode(m.pqmf, y_mb, batch_size=z.shape[:-2], n_channels=m.n_channels)
        outs = dict(x_mb=x_mb.detach().cpu(), z_params=zp.detach().cpu(), y_mb=y_mb.detach().cpu(), y_raw=y_raw.detach().cpu())
        grads = {}
        if backward:
            torch.autograd.backward([y_raw, y_mb, reg], [cy_raw, cy_mb, torch.ones((), device=dev)])
            grads = {k: p.grad.detach().cpu() for k, p in m.named_parameters()
                     if p.grad is not None and k.startswith(("encoder.", "decoder."))}
        m.release_weights()
        torch.cuda.synchronize()
        return outs, grads

    for b in args.batches:
        res[f"b{b}"] = hot_path(b, not args.forward_only)

    # forward-only leg, as bench.py times it
    x = _inputs(32)[0].to(dev)

    def fwd_once():
        with torch.no_grad():
            m.prepare_weights(reuse=True)
            m.decode(m.encoder.reparametrize(m.encode(x))[0])
            m.release_weights()

    for _ in range(3):
        fwd_once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fwd_once()
    e1.record()
    torch.cuda.synchronize()
    res["forward_ms"] = e0.elapsed_time(e1) / 10
    torch.save(res, args.out)


def run_worker(products, out, batches, forward_only):
    from rave_amd import build as B
    env = dict(os.environ)
    if products != 6:
        lib = B.variant_lib(products)
        if not os.path.exists(lib):
            raise SystemExit(f"{lib} missing: python -m rave_amd.build --variants")
        env["RAVE_HIP_LIB"] = lib
    else:
        env.pop("RAVE_HIP_LIB", None)
    cmd = [sys.executable, os.path.abspath(__file__), "--worker", "--out", out, "--batches"] + [str(b) for b in batches]
    if forward_only:
        cmd.append("--forward-only")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    if r.returncode != 0:
        raise SystemExit(f"worker (products {products}) failed:\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}")


def oracle_b8():
    """fp32 outputs and fp64 gradients of the CPU oracle at batch 8 (same seeded model / inputs as the workers)."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import rave_oracle as O
    m = _seeded_model()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    cfg = O.v2_config()
    x, eps, cy_raw, cy_mb = _inputs(8)
    with torch.no_grad():
        out32 = O.rave_forward(x, sd, cfg, eps)
    sd64 = {k: (v.double().requires_grad_(k.startswith(("encoder.", "decoder."))) if v.is_floating_point() else v)
            for k, v in sd.items()}
    out64 = O.rave_forward(x.double(), sd64, cfg, eps.double())
    torch.autograd.backward([out64["y_raw"], out64["y_mb"], out64["reg"]],
                            [cy_raw.double(), cy_mb.double(), torch.ones((), dtype=torch.float64)])
    g64 = {k: v.grad for k, v in sd64.items() if torch.is_tensor(v) and v.is_floating_point() and v.grad is not None}
    return out32, g64


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worker", action="store_true")
    ap.add_argument("--out")
    ap.add_argument("--batches", type=int, nargs="*", default=[8, 32])
    ap.add_argument("--forward-only", action="store_true")
    ap.add_argument("--json", action="store_true", help="bench.py leg: batch-32 forward only, one JSON line")
    ap.add_argument("--md", default=None, help="write the full table here")
    args = ap.parse_args()
    if args.worker:
        return worker(args)
    import torch
    tmp = tempfile.mkdtemp(prefix="x6p_")
    batches = [32] if args.json else [8, 32]
    res = {}
    for n in (6, 4, 3):
        out = os.path.join(tmp, f"p{n}.pt")
        run_worker(n, out, batches, args.json)
        res[n] = torch.load(out, weights_only=False)
        os.remove(out)
    HBM = 8e12
    FWD_BYTES = 2.804e9          # algorithmic bytes of the forward leg
    summary = {}
    for n in (4, 3):
        o6, g6 = res[6]["b32"]
        o, g = res[n]["b32"]
        worst_out = max(rel_l2(o[k], o6[k]) for k in o6)
        summary[f"forward_only_x{n}"] = {
            "dtype": f"bf16 x {n} partial products (2-piece split, f32 accumulate) -- NOT the f32 class of the headline",
            "ms": res[n]["forward_ms"], "frac_of_hbm_roofline": FWD_BYTES / (res[n]["forward_ms"] * 1e-3) / HBM,
            "worst_rel_l2_vs_6_products_b32": worst_out}
    summary["forward_only_x6_same_harness_ms"] = res[6]["forward_ms"]
    if args.json:
        print("X6_PRODUCTS " + json.dumps(summary))
        return
    out32, g64 = oracle_b8()
    lines = ["# bf16 partial products of the forward / data-gradient kernels: 6 (product) vs 4 vs 3 (measurement builds)", "",
             "v2, CAPACITY 96, n_signal 65536; rel-L2.  Weight gradients run on six products in every column.", "",
             "| | 6 products | 4 products | 3 products |", "|---|---|---|---|"]
    row = lambda name, f: lines.append(f"| {name} | " + " | ".join(f(n) for n in (6, 4, 3)) + " |")
    row("forward only, batch 32 (ms)", lambda n: f"{res[n]['forward_ms']:.3f}")
    row("... fraction of the HBM roofline (2.804 GB / 8 TB/s)", lambda n: f"{FWD_BYTES / (res[n]['forward_ms'] * 1e-3) / HBM:.3f}")
    for k in ("x_mb", "z_params", "y_mb", "y_raw"):
        row(f"batch 8: {k} vs fp32 oracle", lambda n, k=k: f"{rel_l2(res[n]['b8'][0][k], out32[k]):.2e}")

    def worst_grad(n, kind, ref):
        w, wk = 0.0, ""
        for k, gr in ref.items():
            if k not in res[n]["b8"][1] or not k.endswith(kind):
                continue
            e = rel_l2(res[n]["b8"][1][k], gr)
            if e > w:
                w, wk = e, k
        return f"{w:.2e} ({wk})"
    row("batch 8: worst weight_v gradient vs fp64 oracle", lambda n: worst_grad(n, "weight_v", g64))
    row("batch 8: worst weight_g gradient vs fp64 oracle", lambda n: worst_grad(n, "weight_g", g64))
    ngr = sum(1 for k in g64 if k in res[6]["b8"][1])
    for k in ("x_mb", "z_params", "y_mb", "y_raw"):
        row(f"batch 32: {k} vs 6 products", lambda n, k=k: "-" if n == 6 else f"{rel_l2(res[n]['b32'][0][k], res[6]['b32'][0][k]):.2e}")

    def worst_vs6(n, kind):
        if n == 6:
            return "-"
        w, wk = 0.0, ""
        for k, gr in res[6]["b32"][1].items():
            if not k.endswith(kind):
                continue
            e = rel_l2(res[n]["b32"][1][k], gr)
            if e > w:
                w, wk = e, k
        return f"{w:.2e} ({wk})"
    row("batch 32: worst weight_v gradient vs 6 products", lambda n: worst_vs6(n, "weight_v"))
    row("batch 32: worst weight_g gradient vs 6 products", lambda n: worst_vs6(n, "weight_g"))
    lines += ["", f"({ngr} gradient tensors compared at batch 8.)"]
    text = "\n".join(lines)
    print(text)
    print("X6_PRODUCTS " + json.dumps(summary))
    if args.md:
        with open(args.md, "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
