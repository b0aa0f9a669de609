"""The reference's training loop (exp/train_utils.py:35-110: shuffled batches, zero_grad / forward / loss / backward / step,
then an evaluation pass) on the device path of this library, from PyG-like graphs to numbers:

    graphs --ring lift on host threads (data/utils.py:501-544)--> PackedComplexes in HBM (with per-complex CSRs)
           --PackedLoader(shuffle=True) (data/data_loading.py:84-111)--> index lists per epoch
           --StaticBatch + StaticTrainStep--> one hipGraph replay per 8 optimisation steps, nothing over PCIe but the
             epoch's permutation
           --StaticForward--> predictions of held-out molecules, bit-identical to model(collate(batch))

    python examples/train_shuffled_epochs.py [n_graphs] [epochs]        (needs an MI355X; synthetic molecules, random labels)
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cwn_amd import csr, lifting                                         # noqa: E402
from cwn_amd.models import EmbedSparseCIN                                # noqa: E402
from cwn_amd.packed import PackedLoader                                  # noqa: E402
from cwn_amd.static_batch import StaticBatch                             # noqa: E402
from cwn_amd.static_graph import StaticForward, StaticTrainStep          # noqa: E402
from cwn_amd.synthetic import random_molecule                            # noqa: E402


def molecule_graph(rng, y_of):
    """A PyG-Data-like dict (what convert_graph_dataset_with_rings takes): both directions of every bond, integer atom /
    bond types as features, one regression target."""
    n, bonds = random_molecule(rng, 12, 30)
    und = np.asarray(bonds, dtype=np.int64).reshape(-1, 2)
    bond_type = rng.integers(0, 4, size=(und.shape[0], 1))
    ei = np.concatenate([und, und[:, ::-1]], axis=0).T
    x = rng.integers(0, 28, size=(n, 1))
    return dict(x=torch.from_numpy(x).float(), edge_index=torch.from_numpy(np.ascontiguousarray(ei)),
                edge_attr=torch.from_numpy(np.concatenate([bond_type, bond_type], axis=0)).float(), num_nodes=n,
                y=torch.tensor([[y_of(x, und)]], dtype=torch.float32))


def main():
    n_graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    dev = torch.device('cuda', 0)
    rng = np.random.default_rng(0)
    # a learnable toy target: (number of carbon-like atoms - number of bonds) / 10
    graphs = [molecule_graph(rng, lambda x, und: float((x < 10).sum() - len(und)) / 10.0) for _ in range(n_graphs)]
    t0 = time.perf_counter()
    packed, dimension, _ = lifting.pack_graph_dataset_with_rings(graphs, max_ring_size=6, init_edges=True, init_rings=False,
                                                                 n_threads=0, device=dev, with_csr=True)
    print(f'lifted + packed {n_graphs} graphs in {time.perf_counter() - t0:.2f} s (dimension {dimension})')
    n_train = n_graphs * 7 // 8
    train_idx, test_idx = np.arange(n_train), np.arange(n_train, n_graphs)
    torch.manual_seed(0)
    model = EmbedSparseCIN(28, 4, 1, 4, 128, dropout_rate=0.0, max_dim=2, jump_mode=None, nonlinearity='relu', readout='sum',
                           train_eps=False, final_hidden_multiplier=2, final_readout='sum', init_reduce='sum', embed_edge=True,
                           use_coboundaries=True, graph_norm='bn').to(dev)
    B, S = 128, 8
    loader = PackedLoader(packed, batch_size=B, shuffle=True, indices=train_idx, seed=1)
    sb = StaticBatch(packed, B, slots=S)
    step = StaticTrainStep(model, sb, task_type='regression', lr=1e-3)
    for epoch in range(epochs):
        loader.set_epoch(epoch)
        batches = loader.batches()
        assert sb.fits(batches).all(), 'a molecule beyond a workgroup: route its batch through PackedComplexes.collate'
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        losses = torch.stack(step.run_epoch(batches)[-S:])      # (device tensors: no sync here; S steps per replay, a shorter
                                                                #  captured sequence for the tail of the epoch)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f'epoch {epoch}: {len(batches)} steps in {dt * 1e3:.1f} ms ({dt / len(batches) * 1e3:.3f} ms / step), '
              f'loss of the last batches {float(losses.mean()):.4f}')
    csr.check_errors(dev)
    # evaluation: model(batch) of exp/train_utils.py:93-110, one graph for every batch
    model.eval()
    ev = StaticForward(model, StaticBatch(packed, B))
    err, n = 0.0, 0
    with torch.no_grad():
        for lo in range(0, len(test_idx), B):
            idx = test_idx[lo:lo + B]
            pred = ev.run(idx)[:len(idx)]
            y = packed.collate(idx).y.view(pred.shape)
            err += float((pred - y).abs().sum())
            n += len(idx)
    print(f'held-out MAE {err / n:.4f} over {n} molecules')


if __name__ == '__main__':
    main()
