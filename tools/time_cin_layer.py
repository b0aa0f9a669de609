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
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 128
b = zinc_like_batch(NB, seed=0, device=dev, include_down_adj=True)
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
print('CINConv layer (ZINC-like batch of %d with lower adjacencies, F=128): fused %.1f us, generic %.1f us' % (NB, run(True), run(False)))


# ---- training: forward + backward of the layer (BatchNorm over the entries in training mode), eager launches ----------------
def run_train(fused):
    from cwn_amd import layers
    layers.FUSED_CIN_TRAINING = fused
    conv.train()
    ws = [torch.randn_like(x) for x in xs]

    def step():
        conv.zero_grad(set_to_none=True)
        xin = [x.clone().requires_grad_() for x in xs]
        b.set_xs(xin)
        out = conv(*b.get_all_cochain_params(max_dim=2, include_down_features=True))
        sum((o * w).sum() for o, w in zip(out, ws)).backward()
    try:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            step()
        e1.record()
        torch.cuda.synchronize()
        eager = e0.elapsed_time(e1) * 100.0
        # ... and the same step replayed from a hipGraph: the device's time without the host's
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            step()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        g.replay()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        return eager, e0.elapsed_time(e1) * 50.0
    finally:
        layers.FUSED_CIN_TRAINING = True
        conv.eval()


a, b_ = run_train(True), run_train(False)
print('CINConv layer forward + backward in training mode: fused %.0f us eager / %.0f us replayed, per-entry path %.0f / %.0f us' % (a[0], a[1], b_[0], b_[1]))
