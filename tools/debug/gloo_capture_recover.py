"""What a refused graph capture leaves behind (two gloo ranks on one GPU; run on the GPU box): which calls still fail after
GraphedTrainingStep.capture raised, and what clears the state."""
import os, sys, socket, traceback
import torch, torch.distributed as dist, torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def worker(rank, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import test_ddp_rccl as T
    dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    from rave_amd import ddp, model as M, ops
    x, eps = T._data()
    xs, es = x[2 * rank:2 * rank + 2].to(dev), eps[2 * rank:2 * rank + 2].to(dev)
    m = T._model(dev); ddp.broadcast_module(m); m.configure_optimizers(capturable=True)
    gen = list(m.encoder.parameters()) + list(m.decoder.parameters())
    red, sync = ddp.GradReducer(gen, bucket_mb=0.25), ddp.BufferSync(m)
    try:
        g = M.GraphedTrainingStep(m, xs, inject_eps=True, grad_begin=lambda i: red.begin(), grad_sync=lambda i: red.finish(), before_step=sync.sync)
        g.capture(xs, 0, eps=es)
        print(rank, "captured", flush=True)
    except Exception as e:
        print(rank, "capture refused:", type(e).__name__, str(e)[:200].replace("\n", " "), flush=True)
        print(rank, "".join(traceback.format_exc().splitlines(True)[-12:]), flush=True)
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    def end_capture(st):
        gph = ctypes.c_void_p()
        return hip.hipStreamEndCapture(ctypes.c_void_p(st.cuda_stream), ctypes.byref(gph))
    def status(st):
        v = ctypes.c_int(-1)
        rc = hip.hipStreamIsCapturing(ctypes.c_void_p(st.cuda_stream), ctypes.byref(v))
        return rc, v.value
    cs = getattr(torch.cuda.graph, "default_capture_stream", None)
    print(rank, "capture stream", cs, "status", status(cs) if cs is not None else None, "side", [status(s) for s in ops._SIDE.values()], flush=True)
    if cs is not None:
        print(rank, "hipStreamEndCapture(capture stream) ->", end_capture(cs), "status", status(cs), "side", [status(s) for s in ops._SIDE.values()], flush=True)
        print(rank, "again ->", end_capture(cs), "status", status(cs), flush=True)
    for name, fn in [("is_capturing", lambda: torch.cuda.is_current_stream_capturing()),
                     ("side capturing", lambda: [s.query() for s in ops._SIDE.values()]),
                     ("synchronize", torch.cuda.synchronize),
                     ("zeros", lambda: torch.zeros(1, device=dev)),
                     ("tensor H2D", lambda: torch.tensor([1], device=dev)),
                     ("tensor H2D again", lambda: torch.tensor([1], device=dev)),
                     ("all_reduce", lambda: dist.all_reduce(torch.ones(1, device=dev)))]:
        try:
            print(rank, name, "->", fn(), flush=True)
        except Exception as e:
            print(rank, name, "FAILED:", type(e).__name__, str(e)[:120].replace("\n", " "), flush=True)
    dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=worker, args=(r, port)) for r in range(2)]
    [p.start() for p in ps]; [p.join(600) for p in ps]
