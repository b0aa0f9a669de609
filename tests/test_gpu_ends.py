"""The two ends of a model forward as single launches (csrc/cwn_ends.hip) through the C ABI:

  * FRONT `cwn_embed_front_f32` (EmbedVEWithReduce.forward, mp/layers.py:490-593) against the separate launches it
    replaces -- bit for bit: same tables, same CSR (= entry) order of every sum -- and against a float64 restatement;
  * HEAD `cwn_head_f32` (pool_complex + lin1s + final readout + lin2, mp/nn.py:50-60, mp/molec_models.py:129-156)
    against the same computation in float64 (gate 1e-5 * max(1, |ref|_inf)) and against the unfused path;
  * the properties the reference tests: a complex's prediction does not depend on the rest of the batch
    (mp/test_molec_models.py:11-68), absent dimensions contribute relu(b1) (mp/nn.py:55-56)."""
import pytest
import torch

from tests._product import gate

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _zinc_model(hidden=128, layers=2, readout='sum', final_readout='sum', jump_mode=None, seed=0):
    from cwn_amd.models import EmbedSparseCIN
    torch.manual_seed(seed)
    return EmbedSparseCIN(28, 4, 1, layers, hidden, dropout_rate=0.0, max_dim=2, jump_mode=jump_mode, nonlinearity='relu',
                          readout=readout, final_readout=final_readout, embed_edge=True, use_coboundaries=True,
                          graph_norm='bn').to(DEV).eval()


def _ends(on):
    from cwn_amd import ops

    class _Ctx:
        def __enter__(self):
            self.prev, ops.FUSED_ENDS = ops.FUSED_ENDS, on

        def __exit__(self, *a):
            ops.FUSED_ENDS = self.prev
    return _Ctx()


@pytest.mark.parametrize('kind', ['zinc', 'molhiv', 'zinc_no_edge_table', 'molhiv_no_edge_table'])
def test_front_bit_identical_to_the_launches_it_replaces(kind):
    from cwn_amd.complex import ComplexBatch
    from cwn_amd.layers import EmbedVEWithReduce, InitReduceConv, OGBEmbedVEWithReduce
    from cwn_amd.models import AtomEncoder, BondEncoder
    from cwn_amd.synthetic import molhiv_like_complexes, zinc_like_complexes
    torch.manual_seed(3)
    H = 64 if kind.startswith('molhiv') else 128
    if kind.startswith('molhiv'):
        # (several tables per cell type: cwn_embed_front_f32 is two launches there -- the embeddings, then the reductions from x0)
        front = OGBEmbedVEWithReduce(AtomEncoder(H), BondEncoder(H) if kind == 'molhiv' else None, InitReduceConv('sum')).to(DEV)
        b = ComplexBatch.from_complex_list(molhiv_like_complexes(40, 5, 6), max_dim=2).to(DEV)
        if kind == 'molhiv_no_edge_table':
            b.cochains[1]._x = None
    else:
        e = torch.nn.Embedding(4, H) if kind == 'zinc' else None
        front = EmbedVEWithReduce(torch.nn.Embedding(28, H), e, InitReduceConv('sum')).to(DEV)
        b = ComplexBatch.from_complex_list(zinc_like_complexes(40, 5, 6), max_dim=2).to(DEV)
        if kind == 'zinc_no_edge_table':
            b.cochains[1]._x = None
    with torch.no_grad():
        with _ends(True):
            got = front(*b.get_all_cochain_params(max_dim=2, include_down_features=False))
        with _ends(False):
            want = front(*b.get_all_cochain_params(max_dim=2, include_down_features=False))
    assert len(got) == len(want) == 3
    for d, (g, w) in enumerate(zip(got, want)):
        assert g.shape == w.shape and torch.equal(g, w), (d, (g - w).abs().max().item())
    # and against a float64 restatement of mp/layers.py:509-547
    vt = [w.detach().double().cpu() for w in ([front.v_embed_layer.weight] if not kind.startswith('molhiv')
                                              else [e_.weight for e_ in front.v_embed_layer.atom_embedding_list])]
    ids0 = b.cochains[0].x.long().cpu()
    x0 = sum(vt[c][ids0[:, c]] for c in range(len(vt)))
    bi1, bi2 = b.cochains[1].boundary_index.cpu(), b.cochains[2].boundary_index.cpu()
    red1 = torch.zeros(b.cochains[1].num_cells, H, dtype=torch.float64).index_add_(0, bi1[1], x0[bi1[0]])
    x2 = torch.zeros(b.cochains[2].num_cells, H, dtype=torch.float64).index_add_(0, bi2[1], red1[bi2[0]]) / 2
    gate(got[0], x0, f'{kind} x0')
    gate(got[2], x2, f'{kind} x2')
    if kind.endswith('_no_edge_table'):
        gate(got[1], red1, f'{kind} x1 = reduced')


@pytest.mark.parametrize('kind', ['zinc', 'zinc_no_edge_table'])
def test_front_backward_in_one_launch(kind):
    """cwn_embed_front_bwd_f32 (round 4): the table gradients of EmbedVEWithReduce (mp/layers.py:490-593) from ONE launch --
    against the launches it replaces (halving, two transposed aggregations, two table gradients) and against autograd over a
    float64 restatement of the forward."""
    from cwn_amd import ops
    from cwn_amd.complex import ComplexBatch
    from cwn_amd.layers import EmbedVEWithReduce, InitReduceConv
    from cwn_amd.synthetic import zinc_like_complexes
    torch.manual_seed(5)
    H = 128
    e = torch.nn.Embedding(4, H) if kind == 'zinc' else None
    front = EmbedVEWithReduce(torch.nn.Embedding(28, H), e, InitReduceConv('sum')).to(DEV)
    b = ComplexBatch.from_complex_list(zinc_like_complexes(37, 6, 6), max_dim=2).to(DEV)
    if kind == 'zinc_no_edge_table':
        b.cochains[1]._x = None
    g = torch.Generator().manual_seed(1)
    ws = [torch.randn(b.cochains[d].num_cells, H, generator=g).to(DEV) for d in range(3)]
    calls = []
    orig = ops._front_backward_fused
    ops._front_backward_fused = lambda *a: (lambda r: (calls.append(r is not None), r)[1])(orig(*a))

    def run(fused):
        ops.FUSED_FRONT_BACKWARD = fused
        front.zero_grad(set_to_none=True)
        with _ends(True):
            out = front(*b.get_all_cochain_params(max_dim=2, include_down_features=False))
        sum((o * w).sum() for o, w in zip(out, ws)).backward()
        return [p.grad.detach().clone() for p in front.parameters()]

    try:
        got = run(True)
        assert calls and calls[-1], 'the one-launch backward was not taken'
        want = run(False)
    finally:
        ops.FUSED_FRONT_BACKWARD, ops._front_backward_fused = True, orig
    for a, r in zip(got, want):
        torch.testing.assert_close(a, r, rtol=1e-5, atol=1e-5 * max(1.0, float(r.abs().max())))
    # float64 autograd of mp/layers.py:509-547
    vt = front.v_embed_layer.weight.detach().double().cpu().requires_grad_(True)
    et = None if e is None else front.e_embed_layer.weight.detach().double().cpu().requires_grad_(True)
    x0 = vt[b.cochains[0].x.long().cpu().view(-1)]
    bi1, bi2 = b.cochains[1].boundary_index.cpu(), b.cochains[2].boundary_index.cpu()
    red1 = torch.zeros(b.cochains[1].num_cells, H, dtype=torch.float64).index_add(0, bi1[1], x0[bi1[0]])
    x1 = red1 if et is None else et[b.cochains[1].x.long().cpu().view(-1)]
    x2 = torch.zeros(b.cochains[2].num_cells, H, dtype=torch.float64).index_add(0, bi2[1], red1[bi2[0]]) / 2
    sum((o * w.double().cpu()).sum() for o, w in zip((x0, x1, x2), ws)).backward()
    gate(got[0], vt.grad, f'{kind}: d vertex table, one-launch front backward')
    if et is not None:
        gate(got[1], et.grad, f'{kind}: d edge table, one-launch front backward')


def test_front_reports_an_index_outside_its_table():
    from cwn_amd import csr
    from cwn_amd.complex import ComplexBatch
    from cwn_amd.layers import EmbedVEWithReduce, InitReduceConv
    from cwn_amd.synthetic import zinc_like_complexes
    front = EmbedVEWithReduce(torch.nn.Embedding(28, 64), torch.nn.Embedding(4, 64), InitReduceConv('sum')).to(DEV)
    b = ComplexBatch.from_complex_list(zinc_like_complexes(4, 6, 6), max_dim=2).to(DEV)
    b.cochains[0].x[3, 0] = 28.0
    with torch.no_grad(), pytest.raises(IndexError):
        front(*b.get_all_cochain_params(max_dim=2, include_down_features=False))
    csr._err_flag(torch.device(DEV)).zero_()       # the sticky word is clean again for the next test


def _head_reference(model, xs, b, readout, final_readout):
    """float64 restatement of mp/nn.py:50-60 + mp/molec_models.py:129-156 on the CPU."""
    C = b.num_complexes
    outs = []
    for d in model.readout_dims:
        lin = model.lin1s[d]
        if d < len(xs):
            x = xs[d].double().cpu()
            bv = b.cochains[d].batch.cpu()
            p = torch.zeros(C, x.size(1), dtype=torch.float64).index_add_(0, bv, x)
            if readout == 'mean':
                p = p / torch.bincount(bv, minlength=C).clamp(min=1).double().unsqueeze(1)
        else:
            p = torch.zeros(C, lin.in_features, dtype=torch.float64)
        bias = lin.bias.detach().double().cpu() if lin.bias is not None else 0.0
        outs.append((p, torch.relu(p @ lin.weight.detach().double().cpu().t() + bias)))
    s = sum(h for _, h in outs)
    if final_readout == 'mean':
        s = s / len(outs)
    return [p for p, _ in outs], s @ model.lin2.weight.detach().double().cpu().t() + model.lin2.bias.detach().double().cpu()


@pytest.mark.parametrize('hidden,readout,final_readout,jump', [(128, 'sum', 'sum', None), (64, 'mean', 'sum', None),
                                                              (64, 'sum', 'mean', 'cat'), (128, 'mean', 'mean', None)])
def test_head_vs_float64_and_vs_the_unfused_launches(hidden, readout, final_readout, jump):
    from cwn_amd.synthetic import zinc_like_batch
    model = _zinc_model(hidden, layers=3 if jump else 2, readout=readout, final_readout=final_readout, jump_mode=jump, seed=4)
    b = zinc_like_batch(33, seed=8, device=DEV)
    K = model.lin1s[0].in_features
    g = torch.Generator().manual_seed(9)
    xs = [torch.randn(b.cochains[d].num_cells, K, generator=g).to(DEV) for d in range(3)]
    res = {}
    with torch.no_grad():
        out = model._head_fused(xs, b, True, res)
    assert out is not None
    pooled, want = _head_reference(model, xs, b, readout, final_readout)
    for k in range(3):
        gate(res[f'pool_{k}'], pooled[k], f'pool_{k} ({readout})')
    gate(out, want, f'head out (hidden {hidden}, {readout}/{final_readout}, jump {jump})')


def test_model_forward_fused_ends_vs_separate_launches():
    """The whole EmbedSparseCIN forward with and without the fused ends: same layers in between, so the difference is
    the ends' own (fp32 re-association of the pooled sums and the GEMV order): inside the gate."""
    from cwn_amd.synthetic import zinc_like_batch
    model = _zinc_model(128, layers=2, seed=11)
    outs = {}
    for on in (True, False):
        b = zinc_like_batch(50, seed=12, device=DEV)
        with torch.no_grad(), _ends(on):
            out, res = model(b, include_partial=True)
        outs[on] = (out, res)
    for k in ('layer0_0', 'layer1_1', 'layer1_2'):
        assert torch.equal(outs[True][1][k], outs[False][1][k]), k       # the front is bit-identical, so are the layers
    for k in ('pool_0', 'pool_1', 'pool_2'):
        gate(outs[True][1][k], outs[False][1][k].double(), k)
    gate(outs[True][0], outs[False][0].double(), 'prediction')


def test_head_absent_dimension_and_batch_independence():
    """A batch without 2-cells: dimension 2 contributes relu(b1) (pooled zeros, mp/nn.py:55-56); and the prediction of
    a complex is bit-identical whatever else is in the batch (one workgroup per complex, fixed summation order)."""
    from cwn_amd.complex import ComplexBatch
    from cwn_amd.synthetic import zinc_like_complexes
    model = _zinc_model(64, layers=1, seed=13)
    cxs = zinc_like_complexes(12, 14, 6)
    K = 64
    g = torch.Generator().manual_seed(15)
    feats = [[torch.randn(c.cochains[d].num_cells, K, generator=g) for d in range(c.dimension + 1)] for c in cxs]

    def run(ids):
        b = ComplexBatch.from_complex_list([cxs[i] for i in ids], max_dim=2).to(DEV)
        xs = [torch.cat([feats[i][d] for i in ids if d < len(feats[i])]).to(DEV) for d in range(b.dimension + 1)]
        with torch.no_grad():
            out = model._head_fused(xs, b, False, {})
        assert out is not None
        return out, xs, b

    full, xs, b = run(list(range(12)))
    part, _, _ = run([7, 2, 9])
    assert torch.equal(part, full[[7, 2, 9]])
    # two dimensions only
    with torch.no_grad():
        out2 = model._head_fused(xs[:2], b, False, {})
    _, want2 = _head_reference(model, xs[:2], b, 'sum', 'sum')
    gate(out2, want2, 'head without dimension 2')


@pytest.mark.parametrize('task', ['regression', 'mse_regression', 'bin_classification'])
def test_fused_loss_matches_the_torch_criterion(task):
    """cwn_loss_f32 (value and gradient in one launch) against the criteria of exp/train_utils.py:20-31, incl. the
    sign(0) = 0 of L1Loss and large logits for the BCE."""
    from cwn_amd.train import _LOSSES, fused_loss
    g = torch.Generator().manual_seed(5)
    for n in (1, 128, 1000):
        pred = (torch.randn(n, 1, generator=g) * 4).to(DEV)
        y = (torch.rand(n, 1, generator=g) > 0.5).float().to(DEV) if task == 'bin_classification' else torch.randn(n, 1, generator=g).to(DEV)
        if n > 1:
            pred[0, 0] = y[0, 0]                                   # an exact hit: sign(0) = 0
            pred[1, 0] = 60.0 if task == 'bin_classification' else pred[1, 0]
        p1, p2 = pred.clone().requires_grad_(True), pred.clone().requires_grad_(True)
        l1 = fused_loss(task, p1, y)
        assert l1 is not None
        (l1 * 3.0).backward()
        l2 = _LOSSES[task](p2, y)
        (l2 * 3.0).backward()
        torch.testing.assert_close(l1, l2, rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(p1.grad, p2.grad, rtol=1e-6, atol=1e-8)
    assert fused_loss('classification', pred, y) is None


@pytest.mark.parametrize('hidden,readout,final_readout', [(128, 'sum', 'sum'), (64, 'mean', 'mean')])
def test_training_head_matches_the_unfused_autograd_path(hidden, readout, final_readout):
    """The head with autograd as two launches (cwn_head_f32 leaving pooled / pre-activations / hidden vector,
    cwn_head_bwd_f32) + the weight gradients through cwn_gemm_tn_f32, against the unfused path (segmented reduce,
    grouped GEMMs, torch adds -- each with its own backward): output, dL/dx of every dimension, every parameter
    gradient; and against float64 CPU autograd of the same formulas."""
    from cwn_amd import ops
    from cwn_amd.synthetic import zinc_like_batch
    model = _zinc_model(hidden, layers=1, readout=readout, final_readout=final_readout, seed=21).train()
    b = zinc_like_batch(29, seed=22, device=DEV)
    K = model.lin1s[0].in_features
    g = torch.Generator().manual_seed(23)
    x0 = [torch.randn(b.cochains[d].num_cells, K, generator=g) for d in range(3)]
    w = torch.randn(b.num_complexes, 1, generator=g).to(DEV)
    grads = {}
    for fused in (True, False):
        prev, ops.FUSED_HEAD_TRAINING = ops.FUSED_HEAD_TRAINING, fused
        try:
            model.zero_grad(set_to_none=True)
            xs = [x.clone().to(DEV).requires_grad_(True) for x in x0]
            res = {}
            out = model._head_fused(xs, b, True, res)
            if not fused:
                assert out is None
                with _ends(False):
                    from cwn_amd.models import pool_complex_list
                    pooled = pool_complex_list(xs, b, 2, readout)
                    hs = [torch.relu(model.lin1s[d](pooled[d])) for d in range(3)]
                    sv = hs[0] + hs[1] + hs[2]
                    out = model.lin2(sv / 3 if final_readout == 'mean' else sv)
            (out * w).sum().backward()
            grads[fused] = (out.detach(), [x.grad for x in xs], {n: p.grad.clone() for n, p in model.named_parameters()
                                                                 if p.grad is not None and ('lin1s' in n or 'lin2' in n)})
        finally:
            ops.FUSED_HEAD_TRAINING = prev
    gate(grads[True][0], grads[False][0].double(), 'training head out')
    for d in range(3):
        gate(grads[True][1][d], grads[False][1][d].double(), f'training head dL/dx[{d}]')
    assert set(grads[True][2]) == set(grads[False][2]) and len(grads[True][2]) == 8
    for n in grads[True][2]:
        gate(grads[True][2][n], grads[False][2][n].double(), f'training head dL/d{n}', tol=2e-5)
    # float64 reference of the whole thing
    lin = [(model.lin1s[d].weight.detach().double().cpu().requires_grad_(True), model.lin1s[d].bias.detach().double().cpu().requires_grad_(True)) for d in range(3)]
    w2, b2 = model.lin2.weight.detach().double().cpu().requires_grad_(True), model.lin2.bias.detach().double().cpu().requires_grad_(True)
    xr = [x.double().requires_grad_(True) for x in x0]
    C = b.num_complexes
    hs = []
    for d in range(3):
        bv = b.cochains[d].batch.cpu()
        p = torch.zeros(C, K, dtype=torch.float64).index_add(0, bv, xr[d])
        if readout == 'mean':
            p = p / torch.bincount(bv, minlength=C).clamp(min=1).double().unsqueeze(1)
        hs.append(torch.relu(p @ lin[d][0].t() + lin[d][1]))
    sv = hs[0] + hs[1] + hs[2]
    ref = (sv / 3 if final_readout == 'mean' else sv) @ w2.t() + b2
    (ref * w.double().cpu()).sum().backward()
    gate(grads[True][0], ref.detach(), 'training head out vs float64')
    for d in range(3):
        gate(grads[True][1][d], xr[d].grad, f'training head dL/dx[{d}] vs float64')
        gate(grads[True][2][f'lin1s.{d}.weight'], lin[d][0].grad, f'dL/dlin1s.{d}.weight vs float64', tol=2e-5)
    gate(grads[True][2]['lin2.weight'], w2.grad, 'dL/dlin2.weight vs float64', tol=2e-5)
    gate(grads[True][2]['lin2.bias'], b2.grad, 'dL/dlin2.bias vs float64', tol=2e-5)


@pytest.mark.parametrize('split', ['1', '4'])
def test_head_over_the_blocks_of_a_jumping_knowledge_concatenation(split):
    """jump_mode 'cat' (mp/models.py:222-232) without the concatenation (round 5): cwn_head_f32 reads the layers' outputs block
    by block (cwn_head_dim.x_more) -- forward BIT-identical to the head over torch.cat (the same rows in the same order through
    the same lanes); with the rows of a complex summed by several workgroups ahead of the head launch (pool_partials: the form
    for complexes of thousands of cells) within the gate of the float64 result.  Backward (cwn_head_bwd_f32 with dx_more and
    a row split): every block's gradient = its columns of the gradient of the concatenation; weight gradients alike."""
    from cwn_amd import ops
    from cwn_amd.synthetic import zinc_like_batch
    K0, n_parts, H2, O = 64, 4, 128, 2
    K = K0 * n_parts
    b = zinc_like_batch(24, seed=12, max_ring=6, device=DEV)
    plan, C = b.block_plan(), b.num_complexes
    g = torch.Generator().manual_seed(2)
    blocks = [[torch.randn(b.cochains[d].num_cells, K0, generator=g).to(DEV).requires_grad_(True) for _ in range(n_parts)] for d in range(3)]
    lin1 = [torch.nn.Linear(K, H2, bias=False).to(DEV) for _ in range(3)]       # (JK cat: bias-free lin1s, mp/models.py:176-181)
    lin2 = torch.nn.Linear(H2, O).to(DEV)
    ptrs = [plan.cell_ptr_device(d, torch.device(DEV)) for d in range(3)]
    params = [q for l in lin1 + [lin2] for q in l.parameters()]
    prev = ops.HEAD_POOL_SPLIT
    ops.HEAD_POOL_SPLIT = split
    try:
        def run(xs):
            for t in [x for blk in blocks for x in blk] + params:
                t.grad = None
            out, _ = ops.head_train(xs, ptrs, C, [l.weight for l in lin1], [None] * 3, lin2.weight, lin2.bias, mean_readout=False,
                                    mean_final=False)
            w = torch.linspace(-1, 1, out.numel(), device=DEV).view_as(out)
            (out * w).sum().backward()
            return out.detach().clone(), [[x.grad.clone() for x in blk] for blk in blocks], [q.grad.clone() for q in params]
        out_p, gx_p, gp_p = run([list(blk) for blk in blocks])
        ops.HEAD_POOL_SPLIT = '1'
        cats = [torch.cat([x for x in blk], dim=-1) for blk in blocks]
        out_c, gx_c, gp_c = run(cats)
    finally:
        ops.HEAD_POOL_SPLIT = prev
    assert torch.equal(out_p, out_c)           # (the same bits however many workgroups sum a complex's rows: one order, cwn_hip.h)
    # float64 reference of the concatenated form
    cd = [c.detach().double() for c in cats]
    pooled = [torch.stack([cd[d][int(ptrs[d][c]):int(ptrs[d][c + 1])].sum(0) for c in range(C)]) for d in range(3)]
    ref = sum(torch.relu(pooled[d] @ lin1[d].weight.detach().double().t()) for d in range(3)) @ lin2.weight.detach().double().t() \
        + lin2.bias.detach().double()
    gate(out_p, ref, f'JK head over blocks, pool split {split}: prediction vs float64')
    for d in range(3):
        for q in range(n_parts):
            gate(gx_p[d][q], gx_c[d][q].double(), f'JK head, pool split {split}: dL/d(block {q} of dim {d}) vs the concatenated form')
    for a, c in zip(gp_p, gp_c):
        gate(a, c.double(), f'JK head, pool split {split}: a weight gradient vs the concatenated form')


def test_large_complexes_pool_to_the_same_bits_by_one_workgroup_or_by_many():
    """REDDIT-like complexes (hundreds to thousands of cells: several chunks of CWN_HEAD_CHUNK rows): the head's prediction and
    the pooled vectors are bit-identical whether one workgroup sums a complex chunk by chunk inside the head launch or
    head_pool_kernel's C x P workgroups do (P = 2, 8, 32), and a complex's rows give the same bits alone and inside a batch."""
    from cwn_amd import ops
    from cwn_amd.complex import ComplexBatch
    from cwn_amd.synthetic import reddit_like_complexes
    pool = reddit_like_complexes(6, seed=4, n_lo=150, n_hi=700)
    b = ComplexBatch.from_complex_list(pool, max_dim=2).to(DEV)
    plan, C = b.block_plan(), b.num_complexes
    K, H2 = 128, 64
    g = torch.Generator().manual_seed(5)
    xs = [torch.randn(b.cochains[d].num_cells, K, generator=g).to(DEV) for d in range(3)]
    assert max(int(plan.cell_ptr[d][c + 1] - plan.cell_ptr[d][c]) for d in range(3) for c in range(C)) > 600
    lin1 = [torch.nn.Linear(K, H2).to(DEV) for _ in range(3)]
    lin2 = torch.nn.Linear(H2, 2).to(DEV)
    ptrs = [plan.cell_ptr_device(d, torch.device(DEV)) for d in range(3)]
    prev, outs = ops.HEAD_POOL_SPLIT, {}
    try:
        for split in ('1', '2', '8', '32'):
            ops.HEAD_POOL_SPLIT = split
            with torch.no_grad():
                outs[split] = ops.head(xs, ptrs, C, [l.weight for l in lin1], [l.bias for l in lin1], lin2.weight, lin2.bias,
                                       mean_readout=True, want_pooled=True)
        ops.HEAD_POOL_SPLIT = '8'
        b1 = ComplexBatch.from_complex_list(pool[3:4], max_dim=2).to(DEV)        # one complex alone
        p1 = b1.block_plan()
        lo = [int(plan.cell_ptr[d][3]) for d in range(3)]
        hi = [int(plan.cell_ptr[d][4]) for d in range(3)]
        with torch.no_grad():
            alone = ops.head([xs[d][lo[d]:hi[d]].contiguous() for d in range(3)], [p1.cell_ptr_device(d, torch.device(DEV)) for d in range(3)], 1,
                             [l.weight for l in lin1], [l.bias for l in lin1], lin2.weight, lin2.bias, mean_readout=True)
    finally:
        ops.HEAD_POOL_SPLIT = prev
    for split in ('2', '8', '32'):
        assert torch.equal(outs[split][0], outs['1'][0]), split
        for d in range(3):
            assert torch.equal(outs[split][1][d], outs['1'][1][d]), (split, d)
    assert torch.equal(alone[0], outs['1'][0][3])
    ref = torch.stack([xs[0].double()[int(ptrs[0][c]):int(ptrs[0][c + 1])].mean(0) for c in range(C)])
    gate(outs['8'][1][0], ref, 'pooled vertices of REDDIT-like complexes (8 workgroups per complex) vs float64')


def test_pooling_launch_with_more_complexes_than_its_lds_table_holds():
    """head_pool_kernel finds a slot's complex by a binary search over the `ptr` table -- staged in LDS up to 2048 entries, read
    from global memory beyond.  2300 complexes of 1 - 300 cells (forced split): same bits as the head launch pooling by itself,
    empty complexes and dimensions without cells for some complexes included."""
    from cwn_amd import ops
    g = torch.Generator().manual_seed(9)
    C, K, H2 = 2300, 64, 32
    sizes = [torch.randint(1, 300, (C,), generator=g), torch.randint(0, 200, (C,), generator=g), torch.randint(0, 3, (C,), generator=g) * 140]
    sizes[0][17] = 0
    ptrs = [torch.cat([torch.zeros(1, dtype=torch.long), s.cumsum(0)]).to(DEV) for s in sizes]
    xs = [torch.randn(int(s.sum()), K, generator=g).to(DEV) for s in sizes]
    lin1 = [torch.nn.Linear(K, H2).to(DEV) for _ in range(3)]
    lin2 = torch.nn.Linear(H2, 1).to(DEV)
    prev, outs = ops.HEAD_POOL_SPLIT, {}
    try:
        for split in ('1', '4'):
            ops.HEAD_POOL_SPLIT = split
            with torch.no_grad():
                outs[split] = ops.head(xs, ptrs, C, [l.weight for l in lin1], [l.bias for l in lin1], lin2.weight, lin2.bias,
                                       mean_readout=False, want_pooled=True)
    finally:
        ops.HEAD_POOL_SPLIT = prev
    assert torch.equal(outs['4'][0], outs['1'][0])
    for d in range(3):
        assert torch.equal(outs['4'][1][d], outs['1'][1][d]), d
    ref = torch.zeros(C, K, dtype=torch.float64).index_add_(0, torch.repeat_interleave(torch.arange(C), sizes[1]), xs[1].double().cpu())
    gate(outs['4'][1][1], ref, 'pooled edges of 2300 complexes (ptr table searched in global memory) vs float64')
