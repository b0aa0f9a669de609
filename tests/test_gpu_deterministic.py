"""ops.deterministic(True) (round 5): the training step with every arrival-ordered sum in its ordered form --
weight gradients through per-band partials reduced in band order, BatchNorm statistics through per-band partials, its backward
sums through the column-owning one-launch form, embedding-table gradients through a stable plan keyed on the table row -- gives
torch.equal gradients run after run, eager against hipGraph replay, and the same parameters after several steps; and is still
the same arithmetic (close to the default mode's gradients)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)


def _setup(kind):
    from cwn_amd.complex import ComplexBatch
    from cwn_amd.models import EmbedSparseCIN, OGBEmbedSparseCIN
    from cwn_amd.synthetic import molhiv_like_complexes, zinc_like_complexes
    torch.manual_seed(3)
    if kind == 'zinc':
        model = EmbedSparseCIN(28, 4, 1, 3, 64, dropout_rate=0.0, max_dim=2, jump_mode=None, nonlinearity='relu', readout='sum',
                               train_eps=False, final_hidden_multiplier=2, final_readout='sum', init_reduce='sum', embed_edge=True,
                               use_coboundaries=True, graph_norm='bn')
        bs = [ComplexBatch.from_complex_list(zinc_like_complexes(48, 70 + i, 6), max_dim=2) for i in range(2)]
        task = 'regression'
    else:
        model = OGBEmbedSparseCIN(1, 2, 64, dropout_rate=0.5, max_dim=2, readout='mean', final_readout='sum', init_reduce='sum',
                                  embed_edge=True, use_coboundaries=True, graph_norm='bn')
        bs = [ComplexBatch.from_complex_list(molhiv_like_complexes(64, 80 + i, 6), max_dim=2) for i in range(2)]
        task = 'bin_classification'
    return model.to(DEV), [b.to(DEV) for b in bs], task


@pytest.mark.parametrize('kind', ['zinc', 'molhiv_dropout'])
def test_deterministic_mode_gives_bit_identical_steps(kind):
    from cwn_amd import ops
    from cwn_amd.train import TrainStep
    model0, _, _ = _setup(kind)
    state0 = {k: v.clone() for k, v in model0.state_dict().items()}

    def run(use_graph, n_steps=3):
        model, bs, task = _setup(kind)
        model.load_state_dict(state0)
        ops.dropout_seed(11, DEV)
        ts = TrainStep(model, bs, task_type=task, lr=1e-3, use_graph=use_graph)
        if use_graph:                              # (the capture's warm-up steps draw masks: rewind, as a fresh run would start)
            for i in range(len(bs)):
                ts._graphs[i] = ts._capture(i)
            ops.dropout_seed(11, DEV)
        losses, grads = [], None
        for i in range(n_steps):
            losses.append(float(ts.step(i % len(bs))))
            if i == 0:
                torch.cuda.synchronize()
                grads = ts.bucket.flat.clone()
        torch.cuda.synchronize()
        return losses, grads, torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone()

    ops.deterministic(True)
    try:
        a = run(False)
        b = run(False)
        c = run(True)
    finally:
        ops.deterministic(False)
    d = run(False)                                 # the default mode: the same arithmetic, sums in arrival order
    assert a[0] == b[0] and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]), 'two eager runs differ'
    assert a[0] == c[0] and torch.equal(a[1], c[1]) and torch.equal(a[2], c[2]), 'eager and replayed steps differ'
    assert abs(a[0][0] - d[0][0]) <= 1e-5 * max(1.0, abs(a[0][0]))
    rel = float((a[1] - d[1]).norm() / d[1].norm())
    print(f'[deterministic, {kind}] gradient of step 0, ordered vs arrival-ordered sums: relative L2 {rel:.2e}')
    assert rel < 1e-4
