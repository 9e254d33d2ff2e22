"""Which piece of the discrete (RVQ + spectral discriminator) step cannot be recorded into a hipGraph?  bench.py --config
discrete segfaults inside hipStreamEndCapture with every new kernel switched off.  One component per process:
warm-up eagerly, then torch.cuda.graph(...) around forward + backward, then one replay."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PARTS = ["stft", "stft_rfft", "encodec_net", "spectral_disc", "msd", "rvq", "discrete_reparam", "encoder", "full_disc"]
if len(sys.argv) > 2 and sys.argv[1] == "worker":
    sys.path.insert(0, ROOT)
    import torch
    from functools import partial
    from rave_amd import model as M, discriminator as D, blocks, quantization as Q
    dev = torch.device("cuda:0")
    part = sys.argv[2]
    torch.manual_seed(0)
    x = (torch.randn(8, 1, 65536, device=dev) * 0.1)
    if part == "stft":
        mod = D.spectrogram(2048).to(dev)
        inp = x.clone().requires_grad_(True)
        fn = lambda: torch.view_as_real(mod(inp)).pow(2).mean()
    elif part == "stft_rfft":
        win = torch.hann_window(2048, device=dev)
        inp = x.clone().requires_grad_(True)
        fn = lambda: torch.view_as_real(torch.fft.rfft(inp.reshape(8, -1).unfold(-1, 2048, 512) * win, dim=-1)).pow(2).mean()
    elif part == "encodec_net":
        mod = D.EncodecConvNet(capacity=32, n_channels=1).to(dev)
        inp = torch.randn(8, 2, 257, 125, device=dev, requires_grad=True)
        fn = lambda: sum(f.pow(2).mean() for f in mod(inp))
    elif part == "spectral_disc":
        mod = D.MultiScaleSpectralDiscriminator([4096, 2048, 1024, 512, 256], partial(D.EncodecConvNet, capacity=32), n_channels=1).to(dev)
        inp = x.clone().requires_grad_(True)
        fn = lambda: sum(f.pow(2).mean() for net in mod(inp) for f in net)
    elif part in ("msd", "full_disc", "encoder", "discrete_reparam", "rvq"):
        m = M.build_discrete().to(dev).train()
        m.encoder.enabled.fill_(1)
        if part == "full_disc":
            mod = m.discriminator
            inp = x.clone().requires_grad_(True)
            fn = lambda: sum(f.pow(2).mean() for net in mod(inp) for f in net)
        elif part == "msd":
            mod = m.discriminator.discriminators[0]
            inp = x.clone().requires_grad_(True)
            fn = lambda: sum(f.pow(2).mean() for net in mod(inp) for f in net)
        else:
            from rave_amd.model import _pqmf_encode
            xm = _pqmf_encode(m.pqmf, x).detach()
            if part == "encoder":
                inp = xm.clone().requires_grad_(True)
                fn = lambda: m.encoder(inp).pow(2).mean()
            else:
                z0 = m.encoder(xm).detach()
                m.encoder.reparametrize(z0)                      # k-means init (host work) outside the capture
                m.set_phase_flags_eagerly()
                inp = z0.clone().requires_grad_(True)
                if part == "rvq":
                    fn = lambda: (lambda o: o[0].pow(2).mean() + o[1])(m.encoder.rvq(inp))
                else:
                    fn = lambda: (lambda o: o[0].pow(2).mean() + o[1])(m.encoder.reparametrize(inp))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            inp.grad = None
            fn().backward()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    inp.grad = None
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loss = fn()
        loss.backward()
    g.replay()
    torch.cuda.synchronize()
    print(f"{part}: captured and replayed, loss {float(loss):.6g}", flush=True)
else:
    for p in PARTS:
        r = subprocess.run([sys.executable, "-X", "faulthandler", os.path.abspath(__file__), "worker", p], capture_output=True, text=True, timeout=200)
        last = (r.stdout.strip().splitlines() or ["(no output)"])[-1]
        err = "" if r.returncode == 0 else " | rc %d | %s" % (r.returncode, " / ".join(l.strip() for l in r.stderr.strip().splitlines() if "File" in l or "Error" in l)[-500:])
        print(last + err, flush=True)
