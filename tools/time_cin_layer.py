"""One CINConv layer (mp/layers.py:62-124, message networks of mp/models.py:40-47) on the ZINC-like batch of
128 with lower adjacencies: fused inference path vs the generic gather -> network -> scatter path."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cwn_amd.layers import CINConv
from cwn_amd.synthetic import zinc_like_batch
dev = torch.device('cuda:0'); torch.manual_seed(0); F = 128
net = lambda: torch.nn.Sequential(torch.nn.Linear(2 * F, F), torch.nn.ReLU(), torch.nn.BatchNorm1d(F))
upd = torch.nn.Sequential(torch.nn.Linear(F, F), torch.nn.ReLU(), torch.nn.Linear(F, F), torch.nn.ReLU(), torch.nn.BatchNorm1d(F))
conv = CINConv(F, F, net(), net(), upd, eps=0.1, max_dim=2).to(dev).eval()
b = zinc_like_batch(128, seed=0, device=dev, include_down_adj=True)
xs = [torch.randn(b.cochains[d].num_cells, F, device=dev) for d in range(3)]
b.set_xs(xs); b.prepare(include_down=True)
params = b.get_all_cochain_params(max_dim=2, include_down_features=True)
def run(fused):
    orig = type(conv.mp_levels[0])._fused_plan
    if not fused: type(conv.mp_levels[0])._fused_plan = lambda self, c: None
    try:
        with torch.no_grad():
            for _ in range(5): conv(*params)
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s): conv(*params)
            torch.cuda.current_stream().wait_stream(s)
            with torch.cuda.graph(g): conv(*params)
            g.replay(); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(50): g.replay()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / 50 * 1e6
    finally:
        type(conv.mp_levels[0])._fused_plan = orig
print('CINConv layer (ZINC-128 with lower adjacencies, F=128): fused %.1f us, generic %.1f us' % (run(True), run(False)))
