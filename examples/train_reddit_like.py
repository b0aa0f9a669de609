"""BASELINE configs[4] as the reference trains it (exp/scripts/mpsn-redditb.sh: SparseCIN over clique complexes of dimension 2,
4 layers of 64, no coboundary features, identity norm, jumping knowledge 'cat', sum readout, CrossEntropyLoss) on the device
path: large irregular complexes with hub vertices go through a static batch in mode 'csr' -- the fill rebuilds every slot's CSR
plans on the device inside the captured graph -- so the shuffled epochs of data/data_loading.py:84-111 replay like everything
else; the head reads the four layers' outputs block by block (no concatenation) and sums a complex's thousands of cells with
many workgroups.

    python examples/train_reddit_like.py [n_graphs] [epochs]        (needs an MI355X; synthetic preferential-attachment graphs)
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cwn_amd import csr                                                   # noqa: E402
from cwn_amd.models import SparseCIN                                      # noqa: E402
from cwn_amd.packed import PackedComplexes, PackedLoader                  # noqa: E402
from cwn_amd.static_batch import StaticBatch                              # noqa: E402
from cwn_amd.static_graph import StaticForward, StaticTrainStep           # noqa: E402
from cwn_amd.synthetic import reddit_like_complexes                       # noqa: E402


def main():
    n_graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    dev = torch.device('cuda', 0)
    t0 = time.perf_counter()
    pool = reddit_like_complexes(n_graphs, seed=0, n_lo=100, n_hi=500)
    for c in pool:          # a toy label: is the graph above the median size?
        c.y = torch.tensor([int(c.cochains[0].num_cells > 300)])
    packed = PackedComplexes(pool, dev, max_dim=2, with_csr=True)
    print(f'{n_graphs} clique complexes lifted and packed in {time.perf_counter() - t0:.1f} s')
    n_train = n_graphs * 7 // 8
    torch.manual_seed(0)
    model = SparseCIN(1, 2, 4, 64, dropout_rate=0.0, max_dim=2, jump_mode='cat', readout='sum', use_coboundaries=False,
                      graph_norm='id').to(dev)
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(0.3)          # (no norm layer and degrees in the hundreds: keep the first activations in range)
    B, S = 32, 4
    loader = PackedLoader(packed, batch_size=B, shuffle=True, indices=np.arange(n_train), seed=1)
    sb = StaticBatch(packed, B, slots=S, mode='csr')
    step = StaticTrainStep(model, sb, task_type='classification', lr=1e-3)
    for epoch in range(epochs):
        loader.set_epoch(epoch)
        batches = loader.batches()
        assert sb.fits(batches).all()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        losses = step.run_epoch(batches)             # (S steps per replay, a shorter captured sequence for the epoch's tail)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f'epoch {epoch}: {len(batches)} steps in {dt * 1e3:.1f} ms ({dt / len(batches) * 1e3:.3f} ms / step), '
              f'mean loss {float(torch.stack(losses[:len(batches)]).mean()):.4f}')
    csr.check_errors(dev)
    model.eval()
    ev = StaticForward(model, StaticBatch(packed, B, mode='csr'))
    hit, n = 0, 0
    with torch.no_grad():
        for lo in range(n_train, n_graphs, B):
            idx = np.arange(lo, min(lo + B, n_graphs))
            pred = ev.run(idx)[:len(idx)]
            y = packed.collate(idx).y.view(-1)
            hit += int((pred.argmax(1) == y).sum())
            n += len(idx)
    print(f'held-out accuracy {hit / n:.3f} over {n} graphs')


if __name__ == '__main__':
    main()
