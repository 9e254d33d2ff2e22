"""Builder-side probe (round 5): does hipStreamEndCapture die on graphs with many cross-stream edges?  The discrete step's capture
segfaults inside the runtime although every component records alone (profiles/round4_discrete_capture_bisect.txt), and forking the
two discriminator families of the v2 GAN-phase step onto two streams makes ITS capture die the same way.  This script records N
iterations of {kernel on stream 1; stream 2 waits for stream 1; kernel on stream 2 [; stream 1 waits for stream 2]} with plain
torch ops -- no kernel of this library -- and replays.  usage: capture_fork_repro.py N [join_every_iteration: 0|1] [kernels per fork]"""
import sys
import torch

n = int(sys.argv[1])
join_each = len(sys.argv) > 2 and sys.argv[2] == "1"
per = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda:0")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
x = torch.zeros(1 << 16, device=dev)
y = torch.zeros_like(x)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s1):
    for i in range(n):
        for _ in range(per):
            x.add_(1.0)
        s2.wait_stream(s1)
        with torch.cuda.stream(s2):
            for _ in range(per):
                y.add_(1.0)
        if join_each:
            s1.wait_stream(s2)
    s1.wait_stream(s2)
print(f"captured: {n} forks, join_each={join_each}, {per} kernels per side", flush=True)
g.replay()
torch.cuda.synchronize()
print("replayed:", float(x[0]), float(y[0]), flush=True)
