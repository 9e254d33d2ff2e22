import os, sys, torch, torch.distributed as dist, torch.multiprocessing as mp
def w(rank, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=2, device_id=torch.device("cuda", 0))
        t = torch.full((1024,), float(rank + 1), device="cuda:0")
        dist.all_reduce(t)
        torch.cuda.synchronize()
        print("rank", rank, "ok", float(t[0]), flush=True)
        dist.destroy_process_group()
    except Exception as e:
        print("rank", rank, "FAILED", repr(e)[:500], flush=True)
if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=w, args=(r, 29611)) for r in range(2)]
    [p.start() for p in ps]; [p.join(120) for p in ps]
    print([p.exitcode for p in ps])
