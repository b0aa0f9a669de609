"""One CINppConv layer (mp/layers.py:216-260, 344-427) in training mode, forward + backward, on a ZINC-like batch with
lower adjacencies: the three streams of all dimensions in one grouped GEMM + one aggregation launch (one autograd node)
vs the reference's own sequence (propagate() + the message hooks per level); `proper` = feed_down_attr (a live lower
stream; the default keeps the reference's quirk: lower stream off).  Device time from a replayed hipGraph.
    python tools/time_cinpp_layer.py [batch] [F]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cwn_amd.layers import CINppConv
from cwn_amd.synthetic import zinc_like_batch

dev = torch.device('cuda:0')
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 128
F = int(sys.argv[2]) if len(sys.argv) > 2 else 64
b = zinc_like_batch(NB, seed=0, device=dev, include_down_adj=True)
xs = [torch.randn(b.cochains[d].num_cells, F, device=dev) for d in range(3)]
ws = [torch.randn_like(x) for x in xs]
b.set_xs(xs)
b.prepare(include_down=True)


def measure(proper, fused):
    torch.manual_seed(0)
    conv = CINppConv(F, F, F, None, None, None, None, None, None, max_dim=2, hidden=F, act_module=torch.nn.ReLU,
                     layer_dim=F, use_coboundaries=True, feed_down_attr=proper).to(dev).train()

    def step():
        conv.zero_grad(set_to_none=True)
        b.set_xs([x.clone().requires_grad_() for x in xs])
        params = b.get_all_cochain_params(max_dim=2, include_down_features=proper)
        out = conv(*params) if fused else [conv.mp_levels[d].forward_unfused(params[d]) for d in range(3)]
        sum((o * w).sum() for o, w in zip(out, ws)).backward()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
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
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 50.0


for proper in (False, True):
    a, c = measure(proper, True), measure(proper, False)
    print('CINppConv layer, batch %d, F=%d, %s: forward + backward (training), replayed: fused streams %.0f us, hook path %.0f us'
          % (NB, F, 'feed_down_attr' if proper else 'default (lower stream off)', a, c))
