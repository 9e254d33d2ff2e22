"""GPU probe: numerics and throughput of the bf16x6 (exact-f32 on bf16 MFMA) conv-like GEMM."""
import ctypes as C, os, subprocess, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "bf16x6.so"))
dev = torch.device("cuda:0")
s = torch.cuda.current_stream().cuda_stream
P = C.c_void_p
for (T, Cc, M, N) in [(1, 96, 96, 131072), (3, 96, 96, 131072), (3, 384, 384, 8192), (3, 768, 768, 2048), (1, 768, 768, 2048)]:
    torch.manual_seed(0)
    NX = N + (4 if T > 1 else 0)     # rows stay 16-byte aligned
    x = torch.randn(Cc, NX, device=dev)
    w = torch.randn(T, Cc, M, device=dev) / (T * Cc) ** 0.5
    wq = torch.empty(T * Cc * M * 3, device=dev, dtype=torch.int16)
    out = torch.empty(M, N, device=dev)
    assert lib.split_weights(P(w.data_ptr()), P(wq.data_ptr()), T, Cc, M, P(s)) == 0
    rc = lib.gemm(P(x.data_ptr()), P(wq.data_ptr()), P(out.data_ptr()), T, Cc, M, N, NX, P(s), 0)
    assert rc == 0, rc
    torch.cuda.synchronize()
    ref = sum(w[t].double().t() @ x[:, t:t + N].double() for t in range(T))
    ref32 = sum(w[t].t() @ x[:, t:t + N] for t in range(T))
    err = float((out.double() - ref).norm() / ref.norm())
    err32 = float((ref32.double() - ref).norm() / ref.norm())
    fl = 2.0 * T * Cc * M * N
    res = []
    for mode in (0, 1, 2, 3, 4, 7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            lib.gemm(P(x.data_ptr()), P(wq.data_ptr()), P(out.data_ptr()), T, Cc, M, N, NX, P(s), mode)
        e0.record()
        for _ in range(10):
            lib.gemm(P(x.data_ptr()), P(wq.data_ptr()), P(out.data_ptr()), T, Cc, M, N, NX, P(s), mode)
        e1.record()
        torch.cuda.synchronize()
        res.append((mode, e0.elapsed_time(e1) / 10))
    print(f"T={T} C={Cc} M={M} N={N}: rel err vs f64 {err:.2e} (torch f32 matmul {err32:.2e}); " +
          ", ".join(f"mode{m}: {ms*1e3:.0f}us" for m, ms in res) + f" -> {fl/res[0][1]/1e9:.1f} TF/s")
    out2 = torch.empty(M, N, device=dev)
    rc = lib.gemm_v2(P(x.data_ptr()), P(wq.data_ptr()), P(out2.data_ptr()), T, Cc, M, N, NX, P(s), 0)
    assert rc == 0, rc
    torch.cuda.synchronize()
    err2 = float((out2.double() - ref).norm() / ref.norm())
    res2 = []
    for mode in (0, 1, 2, 3, 4, 7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            lib.gemm_v2(P(x.data_ptr()), P(wq.data_ptr()), P(out2.data_ptr()), T, Cc, M, N, NX, P(s), mode)
        e0.record()
        for _ in range(10):
            lib.gemm_v2(P(x.data_ptr()), P(wq.data_ptr()), P(out2.data_ptr()), T, Cc, M, N, NX, P(s), mode)
        e1.record()
        torch.cuda.synchronize()
        res2.append((mode, e0.elapsed_time(e1) / 10))
    print(f"   v2: rel err {err2:.2e}; " + ", ".join(f"mode{m}: {ms*1e3:.0f}us" for m, ms in res2) + f" -> {fl/res2[0][1]/1e9:.1f} TF/s")
