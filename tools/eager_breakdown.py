"""Where the host time of an eagerly launched propagate-scope step goes (no graph replay)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cwn_amd import csr, ops
from cwn_amd.models import EmbedSparseCIN
from cwn_amd.synthetic import zinc_like_batch
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = EmbedSparseCIN(28, 4, 1, 4, 128, dropout_rate=0.0, embed_edge=True, use_coboundaries=True).to(dev).eval()
b = zinc_like_batch(128, seed=0, device=dev)
feats = [torch.randn(b.cochains[d].num_cells, 128, device=dev) for d in range(3)]
T = {}
def tick(name, t0):
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0
def step():
    t = time.perf_counter(); csr._cache.clear(); b.prepare(max_dim=2); tick('prepare (plan build call)', t)
    for conv in model.convs:
        t = time.perf_counter(); b.set_xs(feats); params = b.get_all_cochain_params(max_dim=2, include_down_features=False); tick('set_xs + get_all_cochain_params', t)
        t = time.perf_counter()
        specs, owner = [], []
        for dim in range(3):
            sp = conv.mp_levels[dim].gemm_specs(params[dim]); specs += sp; owner += [dim] * len(sp)
        tick('gemm_specs', t)
        t = time.perf_counter(); ys = ops.gemm_many(specs); tick('gemm_many (launch)', t)
        t = time.perf_counter()
        plans = [conv.mp_levels[dim].streams(params[dim], [y for y, o in zip(ys, owner) if o == dim] or None) for dim in range(3)]
        tick('streams()', t)
        t = time.perf_counter(); ops.aggregate_many([st for p in plans for st in p]); tick('aggregate_many (launch)', t)
with torch.no_grad():
    for _ in range(20): step()
    torch.cuda.synchronize(); T.clear()
    t0 = time.perf_counter()
    for _ in range(200): step()
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / 200 * 1e6
print(f'total {total:.0f} us/step (host-bound)')
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
    print(f'  {v / 200 * 1e6:7.1f} us  {k}')
