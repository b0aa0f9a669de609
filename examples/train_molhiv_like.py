"""BASELINE configs[2] as the reference trains it (exp/scripts/cwn-molhiv.sh: OGBEmbedSparseCIN, 2 layers of 64, mean readout,
dropout 0.5 after every conv layer and before lin2, BCE with logits, lr 1e-4) on the device path of this library, on
molhiv-LIKE molecules that include the dataset's heavy tail (a few molecules of 120 - 220 atoms: beyond what one workgroup of
the blocked layer kernel holds):

    complexes --> PackedComplexes in HBM --> PackedLoader(shuffle=True): index lists per epoch
              --> StaticRouter: every epoch split between the blocked static batch (molecules that fit) and the csr-mode one
              --> RoutedTrainStep: two captured graphs over ONE model / gradient bucket / Adam state; dropout multipliers are
                  derived inside the kernels (no mask tensors), fresh per replayed step
              --> RoutedForward: the evaluation pass (eval mode: dropout off)

    python examples/train_molhiv_like.py [n_molecules] [epochs] [tail]     (needs an MI355X; synthetic molecules, a toy label;
    tail = the share of 120 - 220-atom molecules, default 2e-3)
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cwn_amd import csr                                                   # noqa: E402
from cwn_amd.models import OGBEmbedSparseCIN                              # noqa: E402
from cwn_amd.packed import PackedComplexes, PackedLoader                  # noqa: E402
from cwn_amd.static_graph import RoutedForward, RoutedTrainStep, StaticRouter   # noqa: E402
from cwn_amd.synthetic import molhiv_like_complexes                       # noqa: E402


def main():
    n_mol = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    tail = float(sys.argv[3]) if len(sys.argv) > 3 else 2e-3
    dev = torch.device('cuda', 0)
    t0 = time.perf_counter()
    pool = molhiv_like_complexes(n_mol, seed=0, max_ring=6, tail=tail)
    for c in pool:          # a toy label a model can learn: does the molecule have more than two rings?
        c.y = torch.tensor([[float(c.cochains[2].num_cells > 2 if c.dimension >= 2 else 0.0)]])
    packed = PackedComplexes(pool, dev, max_dim=2, with_csr=True)
    big = sum(c.cochains[0].num_cells > 100 for c in pool)
    print(f'{n_mol} molecules ({big} of more than 100 atoms) lifted and packed in {time.perf_counter() - t0:.1f} s')
    n_train = n_mol * 7 // 8
    torch.manual_seed(0)
    model = OGBEmbedSparseCIN(1, 2, 64, dropout_rate=0.5, indropout_rate=0.0, max_dim=2, readout='mean', final_readout='sum',
                              apply_dropout_before='lin2', init_reduce='sum', embed_edge=True, use_coboundaries=True,
                              graph_norm='bn').to(dev)
    B, S = 128, 8
    loader = PackedLoader(packed, batch_size=B, shuffle=True, indices=np.arange(n_train), seed=1)
    step = RoutedTrainStep(model, StaticRouter(packed, B, slots=S), task_type='bin_classification', lr=1e-4)
    for epoch in range(epochs):
        loader.set_epoch(epoch)
        batches = loader.batches()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        losses = step.run_epoch(batches)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        a, b = step.router.split(batches)
        print(f'epoch {epoch}: {len(batches)} steps in {dt * 1e3:.1f} ms ({dt / len(batches) * 1e3:.3f} ms / step; {len(a)} batches on the '
              f'blocked path, {len(b)} on the streaming path), mean loss {float(torch.stack(losses).mean()):.4f}')
    csr.check_errors(dev)
    model.eval()
    ev = RoutedForward(model, StaticRouter(packed, B, slots=S))
    test = [np.arange(lo, min(lo + B, n_mol)) for lo in range(n_train, n_mol, B)]
    hit, n = 0, 0
    with torch.no_grad():
        for idx, pred in zip(test, ev.run_epoch(test)):
            y = packed.collate(idx).y.view(pred.shape)
            hit += int(((pred > 0).float() == y).sum())
            n += len(idx)
    print(f'held-out accuracy {hit / n:.3f} over {n} molecules (Adam steps taken: {int(step.opt.t)})')


if __name__ == '__main__':
    main()
