"""Eager training step (forward, L1 loss, backward, Adam) of EmbedSparseCIN on a ZINC-like batch:
wall time per step and, under rocprofv3, the kernel mix."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cwn_amd.models import EmbedSparseCIN
from cwn_amd.synthetic import zinc_like_batch

dev = torch.device('cuda:0')
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
model = EmbedSparseCIN(28, 4, 1, 4, 128, dropout_rate=0.0, embed_edge=True, use_coboundaries=True).to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
batches = [zinc_like_batch(B, seed=i, device=dev) for i in range(4)]
types = [(b.cochains[0].x.clone(), b.cochains[1].x.clone()) for b in batches]
for b in batches:
    b.prepare(backward=True)


def step(i):
    b = batches[i % 4]
    b.cochains[0]._x, b.cochains[1]._x, b.cochains[2]._x = types[i % 4][0], types[i % 4][1], None
    opt.zero_grad(set_to_none=True)
    y = model(b)
    loss = (y - b.y.view(-1, 1)).abs().mean()
    loss.backward()
    opt.step()
    return loss


for i in range(5):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    loss = step(i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
cells = sum(int(batches[0].cochains[d].num_cells) for d in range(3))
print(f'train step {dt * 1e3:.3f} ms (eager), loss {float(loss):.4f}, {cells * 4 / dt / 1e6:.2f} M cells/s')
