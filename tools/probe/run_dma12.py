import ctypes, torch, os
lib = ctypes.CDLL(os.path.join(os.path.dirname(__file__), "dma12.so"))
x = torch.arange(4096, dtype=torch.float32, device="cuda")
out = torch.zeros(256, device="cuda")
rc = lib.run(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(out.data_ptr()), 4096, 1, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
o = out.cpu().tolist()
print("rc", rc)
print("lds[0:12]  ", o[0:12])
print("lds[3:3+24]", o[3:27])
exp = []
for lane in range(64):
    base = lane * 5 + 1
    exp += [0, 0, 0] if lane == 7 else [base, base + 1, base + 2]
print("match contiguous 12B/lane layout:", o[3:3 + 192] == [float(v) for v in exp])
