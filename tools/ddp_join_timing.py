"""What does joining the weight-gradient side stream at every bucket boundary cost while a REAL collective is in flight?
(`GradReducer._launch` joins the side stream before a bucket leaves; measured so far only with one
rank, where the all-reduce is a no-op.)

A gpurun box has ONE MI355X and RCCL refuses two ranks on one device, so two ranks share the GPU and exchange their CUDA
gradient buckets through gloo (host staging: far SLOWER than xGMI -- the collective is in flight for longer than it would be on
an 8-GPU node, which makes this an upper bound on what the joins can cost).  Full-width v2 (CAPACITY 96), batch 8 per rank,
eager VAE-phase steps.  Per configuration: ms per step (max over ranks), the time the compute stream spends in
`GradReducer.finish()` (= communication not hidden behind backward), the overlapped byte fraction.

  side=1  weight-gradient branch on its side stream, joined at each of the bucket boundaries (the product default)
  side=0  no side stream: nothing to join (hooks launch the buckets straight from the compute stream)

Run: python tools/ddp_join_timing.py [--steps 12] [--batch 8]      (spawns the two ranks itself)"""
import argparse
import json
import os
import socket
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(rank, world, port, args, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rave_amd import ddp, model as M
    torch.manual_seed(0)
    m = M.build_v2().to(dev).train()
    ddp.broadcast_module(m)
    m.configure_optimizers()
    gen = list(m.encoder.parameters()) + list(m.decoder.parameters())
    red = ddp.GradReducer(gen, force=world == 1)
    g = torch.Generator().manual_seed(100 + rank)
    x = (0.1 * torch.randn(args.batch, 1, 65536, generator=g)).clamp(-1, 1).to(dev)
    res = {}
    for side in (1, 0, 1, 0):
        os.environ["RH_BWD_SIDE_STREAM"] = str(side)
        exposed = []

        def begin(_i):
            red.begin()

        def sync(_i):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            red.finish()
            e1.record()
            exposed.append((e0, e1, time.perf_counter() - t0))

        for i in range(3):
            m.training_step(x.detach().clone(), i, grad_begin=begin, grad_sync=sync)
        torch.cuda.synchronize()
        dist.barrier()
        exposed.clear()
        red.bytes_reduced = red.bytes_overlapped = 0
        t0 = time.perf_counter()
        for i in range(args.steps):
            m.training_step(x.detach().clone(), 3 + i, grad_begin=begin, grad_sync=sync)
        torch.cuda.synchronize()
        dt = torch.tensor([time.perf_counter() - t0])
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        key = f"side={side}" + ("" if f"side={side}" not in res else " (second run)")
        res[key] = {"ms_per_step": float(dt) / args.steps * 1e3,
                    "finish_gpu_ms": sum(a.elapsed_time(b) for a, b, _ in exposed) / len(exposed),
                    "finish_host_ms": sum(c for _, _, c in exposed) / len(exposed) * 1e3,
                    "overlapped_fraction": red.bytes_overlapped / max(red.bytes_reduced, 1),
                    "buckets": len(red.buckets)}
    if rank == 0:
        with open(out, "w") as f:
            json.dump(res, f)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--world", type=int, default=2, help="1 = a single rank (the gloo staging path with nobody to talk to)")
    args = ap.parse_args()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = os.path.join("/tmp", f"ddp_join_{os.getpid()}.json")
    mp.spawn(_worker, args=(args.world, port, args, out), nprocs=args.world, join=True)
    res = json.load(open(out))
    print(f"{args.world} rank(s) on one MI355X over gloo, v2 CAPACITY 96, batch {args.batch} per rank, eager VAE-phase step, {args.steps} steps")
    print(f"{'configuration':24s} {'ms/step':>9s} {'finish() GPU ms':>16s} {'finish() host ms':>17s} {'overlapped':>11s} {'buckets':>8s}")
    for k, v in res.items():
        print(f"{k:24s} {v['ms_per_step']:9.2f} {v['finish_gpu_ms']:16.3f} {v['finish_host_ms']:17.3f} {v['overlapped_fraction']:11.2f} {v['buckets']:8d}")


if __name__ == "__main__":
    main()
