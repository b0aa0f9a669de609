"""Whole training step (forward, L1 loss, backward, Adam) of EmbedSparseCIN captured in ONE hipGraph
per batch and replayed; compares with the eager step.  usage: train_graph.py [batch] [steps]"""
import faulthandler
import os
import sys
import time
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cwn_amd.models import EmbedSparseCIN
from cwn_amd.synthetic import zinc_like_batch

dev = torch.device('cuda:0')
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
model = EmbedSparseCIN(28, 4, 1, 4, 128, dropout_rate=0.0, embed_edge=True, use_coboundaries=True).to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True, fused=os.environ.get('CWN_FUSED_ADAM', '1') == '1')
NB = 4
batches = [zinc_like_batch(B, seed=i, device=dev) for i in range(NB)]
types = [(b.cochains[0].x.clone(), b.cochains[1].x.clone()) for b in batches]
for b in batches:
    b.prepare(backward=True)
torch.cuda.synchronize()


def step(i):
    b = batches[i % NB]
    b.cochains[0]._x, b.cochains[1]._x, b.cochains[2]._x = types[i % NB][0], types[i % NB][1], None
    opt.zero_grad(set_to_none=True)
    y = model(b)
    loss = (y - b.y.view(-1, 1)).abs().mean()
    loss.backward()
    # the model leaves its last layer's outputs in the batch (set_xs, as the reference does):
    # drop them, they hold the autograd graph
    b.cochains[0]._x, b.cochains[1]._x, b.cochains[2]._x = types[i % NB][0], types[i % NB][1], None
    opt.step()
    return loss.detach()     # keep no reference to the autograd graph (its AccumulateGrad nodes
                             # would stay bound to the stream of THIS iteration)


def timeit(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        out = fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, out


for i in range(5):
    step(i)
dt_eager, loss = timeit(step, steps)
cells = sum(int(batches[0].cochains[d].num_cells) for d in range(3))
print(f'eager   : {dt_eager * 1e3:8.3f} ms/step  loss {float(loss):.4f}  {cells * 4 / dt_eager / 1e6:8.2f} M cells/s', flush=True)

side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for i in range(NB):
        step(i)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
graphs, losses = [], []
for i in range(NB):
    g = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    print(f'capturing batch {i}', flush=True)
    with torch.cuda.graph(g):
        losses.append(step(i))
    graphs.append(g)
torch.cuda.synchronize()
print('captured', flush=True)


def replay(i):
    graphs[i % NB].replay()
    return losses[i % NB]


for i in range(5):
    replay(i)
dt_graph, loss = timeit(replay, steps)
print(f'hipGraph: {dt_graph * 1e3:8.3f} ms/step  loss {float(loss):.4f}  {cells * 4 / dt_graph / 1e6:8.2f} M cells/s')
