"""The full forward of the molhiv-like (or ZINC-like) configuration: fixed batches replayed from a graph against a blocked static
batch over the SAME batches (run under rocprofv3: tools/prof_static_forward.sh) -- which launches does the static form run slower?
usage: prof_static_forward.py fixed|static [molhiv|zinc]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cwn_amd.complex import ComplexBatch
from cwn_amd.models import EmbedSparseCIN, OGBEmbedSparseCIN
from cwn_amd.packed import PackedComplexes
from cwn_amd.static_batch import StaticBatch
from cwn_amd.static_graph import StaticForward
from cwn_amd.synthetic import molhiv_like_complexes, zinc_like_complexes
dev = torch.device('cuda', 0)
which, wl = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else 'molhiv')
torch.manual_seed(0)
if wl == 'molhiv':
    model = OGBEmbedSparseCIN(1, 2, 64, dropout_rate=0.0, max_dim=2, readout='mean', final_readout='sum', init_reduce='sum',
                              embed_edge=True, use_coboundaries=True, graph_norm='bn').to(dev).eval()
    B, gen = 512, (lambda s: molhiv_like_complexes(512, s, 6))
else:
    model = EmbedSparseCIN(28, 4, 1, 4, 128, dropout_rate=0.0, embed_edge=True, use_coboundaries=True).to(dev).eval()
    B, gen = 128, (lambda s: zinc_like_complexes(128, s, 6))
S = 4
pool = [c for i in range(S) for c in gen(70 + i)]
with torch.no_grad():
    if which == 'fixed':
        bs = [ComplexBatch.from_complex_list(pool[i * B:(i + 1) * B], max_dim=2).to(dev) for i in range(S)]
        xs = [[b.cochains[d].x for d in range(3)] for b in bs]
        graphs = []
        for b, x0 in zip(bs, xs):
            for _ in range(2):
                b.set_xs(x0); model(b)
            g = torch.cuda.CUDAGraph()
            b.set_xs(x0)
            with torch.cuda.graph(g):
                out = model(b)
            b.set_xs(x0)
            graphs.append((g, out))
        for r in range(20):
            for g, _ in graphs:
                g.replay()
    else:
        packed = PackedComplexes(pool, dev, max_dim=2, with_csr=True)
        sb = StaticBatch(packed, B, slots=S)
        sf = StaticForward(model, sb)
        perm = np.arange(len(pool))
        for r in range(20):
            sf.run_epoch([perm[k * B:(k + 1) * B] for k in range(S)])
torch.cuda.synchronize()
print('done', which, wl)
