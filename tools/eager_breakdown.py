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
    t = time.perf_counter(); b.block_plan().forget_csr(); tick('forget_csr', t)
    for conv in model.convs:
        t = time.perf_counter(); b.set_xs(feats); params = b.get_all_cochain_params(max_dim=2, include_down_features=False); tick('set_xs + get_all_cochain_params', t)
        t = time.perf_counter(); conv._propagate_blocked(params, 0); tick('_propagate_blocked', t)
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
import cProfile, pstats
pr = cProfile.Profile()
with torch.no_grad():
    pr.enable()
    for _ in range(100): step()
    pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
