"""The whole optimisation step (cwn_amd.train.TrainStep) on ZINC-like batches, graph-captured vs
eager; under rocprofv3 --kernel-trace --stats the kernel mix of a step.
usage: train_graph.py [batch] [steps] [workload] [dropout]   (molhiv: BCE with logits, dropout default 0.5 = exp/scripts/cwn-molhiv.sh)"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cwn_amd.complex import ComplexBatch
from cwn_amd.models import EmbedSparseCIN, OGBEmbedSparseCIN, SparseCIN
from cwn_amd.synthetic import zinc_like_complexes, molhiv_like_complexes, reddit_like_complexes, batch_stats
from cwn_amd.train import TrainStep

dev = torch.device('cuda:0')
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
WL = sys.argv[3] if len(sys.argv) > 3 else 'zinc'
DROP = float(sys.argv[4]) if len(sys.argv) > 4 else (0.5 if WL != 'zinc' else 0.0)
if WL == 'zinc':
    model = EmbedSparseCIN(28, 4, 1, 4, 128, dropout_rate=0.0, embed_edge=True, use_coboundaries=True).to(dev)
    gen = lambda s: zinc_like_complexes(B, s, 6)
elif WL == 'reddit':       # exp/scripts/mpsn-redditb.sh: SparseCIN 64 x 4, no coboundaries, identity norm, JK cat, cross-entropy
    model = SparseCIN(1, 2, 4, 64, dropout_rate=DROP if len(sys.argv) > 4 else 0.0, max_dim=2, jump_mode='cat', readout='sum',
                      use_coboundaries=False, graph_norm='id').to(dev)
    with torch.no_grad():
        for p_ in model.parameters():
            p_.mul_(0.3)
    gen = lambda s: reddit_like_complexes(B, s)
else:
    model = OGBEmbedSparseCIN(1, 2, 64, dropout_rate=DROP, max_dim=2, readout='mean', final_readout='sum',
                              init_reduce='sum', embed_edge=True, use_coboundaries=True, graph_norm='bn').to(dev)
    gen = lambda s: molhiv_like_complexes(B, s, 6)
batches = [ComplexBatch.from_complex_list(gen(i), max_dim=2).to(dev) for i in range(2)]
for b in batches:
    if WL == 'reddit':
        b.y = torch.zeros(b.num_complexes, dtype=torch.long, device=dev)
    elif b.y is None or WL != 'zinc':
        b.y = torch.zeros(b.num_complexes, 1, device=dev)
cells = batch_stats(batches[0])['cells'] * len(model.convs)
for graph in ([False, True] if os.environ.get('EAGER_TOO') else [True]):
    ts = TrainStep(model, batches, task_type={'zinc': 'regression', 'reddit': 'classification'}.get(WL, 'bin_classification'), use_graph=graph)
    for i in range(4):
        ts.step(i % 2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = ts.step(i % 2)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f'{"hipGraph" if graph else "eager   "}: {dt * 1e3:8.3f} ms/step  loss {float(loss):.4f}  {cells / dt / 1e6:8.2f} M cells/s')
