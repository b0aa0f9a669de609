"""Throughput of the propagate scope on a molecular batch with a few OVERSIZED complexes: all-small
batch vs the same batch with 5 % of its molecules replaced by giants of 60 - 200 atoms, (a) giants streamed inside the
blocked launch (BIG records), (b) the round-2 behaviour: the whole batch on the two-kernel path.
usage: big_items_bench.py [zinc|molhiv] [batch] [share]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cwn_amd import csr, layers
from cwn_amd.complex import ComplexBatch
from cwn_amd.layers import SparseCINConv
from cwn_amd.synthetic import zinc_like_complexes, molhiv_like_complexes, batch_stats

WL = sys.argv[1] if len(sys.argv) > 1 else 'molhiv'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
share = float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
F, L = (64, 2) if WL == 'molhiv' else (128, 4)
dev = torch.device('cuda:0')
gen = (lambda n, s, **kw: molhiv_like_complexes(n, s, 6, **kw)) if WL == 'molhiv' else (lambda n, s, **kw: zinc_like_complexes(n, s, 6, **kw))
small = gen(B, 0)
n_big = max(1, int(round(share * B)))
giants = gen(n_big, 1, n_lo=60, n_hi=200)
mixed = list(small)
for k, gx in enumerate(giants):
    mixed[(k * 37 + 5) % B] = gx
torch.manual_seed(0)
convs = [SparseCINConv(F, F, F, None, None, None, None, max_dim=2, hidden=F, act_module=torch.nn.ReLU, layer_dim=F,
                       use_coboundaries=True).to(dev).eval() for _ in range(L)]


def measure(cxs, big_items, blocked=True):
    layers.BIG_ITEMS, layers.BLOCKED_LAYER = big_items, blocked
    layers._BLOCKED_CACHE.clear()
    csr._cache.clear()
    b = ComplexBatch.from_complex_list(cxs, max_dim=2).to(dev)
    for d in range(3):
        b.cochains[d].x = torch.randn(b.cochains[d].num_cells, F, device=dev)
    cells = batch_stats(b)['cells'] * L

    def step():
        if blocked and b.block_plan() is not None:
            b.block_plan().forget_csr()
        if not blocked or getattr(convs[0], 'blocked_reason', None) is not None:
            csr._cache.clear()
            b.prepare(max_dim=2)
        for conv in convs:
            params = b.get_all_cochain_params(max_dim=2, include_down_features=False)
            plans, outs = conv.propagate_all(*params)
        return plans, outs
    with torch.no_grad():
        plans, _ = step()
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            step()
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            keep = [step() for _ in range(10)]
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            g.replay()
        e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 100
    return plans[0], us, cells / us


rows = [('all small', small, True, True), (f'{n_big} giants of 60-200 atoms, streamed in the blocked launch', mixed, 'always', True),
        ('same batch, library policy (layers._streaming_pays)', mixed, True, True),
        ('same batch, whole batch on the two-kernel path (round 2)', mixed, False, True),
        ('all small, two-kernel path', small, True, False)]
base = None
for name, cxs, big, blocked in rows:
    plan, us, rate = measure(cxs, big, blocked)
    base = base or rate
    print(f'{WL}-{B} {name}: path {plan}, {us:.1f} us/step, {rate:.1f} M cells/s ({rate / base:.2f} of all-small)')
