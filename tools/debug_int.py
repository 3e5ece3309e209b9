import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import minnow_amd as mnw
dev = torch.device("cuda:0")
N, B, bits = 5 * 10**7, 2**22, 16
ctx = mnw.Context(0)
g = torch.Generator(device=dev); g.manual_seed(bits)
x = 1000 + torch.randint(0, 2**bits, (N,), device=dev, dtype=torch.int64, generator=g)
nblk = (N + B - 1) // B
slot = 8 * B
packed = torch.zeros(nblk * slot, dtype=torch.uint8, device=dev)
ej = [mnw.EncJob(x.data_ptr() + 8*k*B, packed.data_ptr() + k*slot, min(B, N-k*B), slot, mnw.INT_GROUP) for k in range(nblk)]
for rep in range(3):
    metas = ctx.encode_blocks(ej)
    for k, m in enumerate(metas):
        xb = x[k*B: k*B + ej[k].n]
        mn, mx = int(xb.min()), int(xb.max())
        flag = "" if (m.min == mn and m.bits == (mx - mn).bit_length()) else "  <-- WRONG (true min %d max %d)" % (mn, mx)
        print(rep, k, ej[k].n, m.min, m.bits, m.nbytes, m.status, flag)
