"""Plan-build time against the length of ONE hub row (general path): the stable in-row rank."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cwn_amd.csr import Adjacency
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
n = 200_000
for hub in (300, 3_000, 30_000, 300_000):
    dst = torch.cat([torch.full((hub,), 17), torch.randint(0, n, (400_000,), generator=g)])
    dst = dst[torch.randperm(dst.numel(), generator=g)]
    src = torch.randint(0, n, (dst.numel(),), generator=g)
    idx = torch.stack([src, dst]).to(dev)
    Adjacency.from_index(idx, n, n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        Adjacency.from_index(idx, n, n)
    torch.cuda.synchronize()
    print(f'hub of {hub:7d} entries: {(time.perf_counter() - t0) / 3 * 1e3:8.3f} ms per plan build', flush=True)
