import sys, os; sys.path.insert(0, '/root/repo')
import torch, numpy as np, faulthandler
faulthandler.enable()
from cwn_amd import csr, _ffi
from cwn_amd.models import EmbedCINpp
from cwn_amd.packed import PackedComplexes, PackedLoader
from cwn_amd.static_batch import StaticBatch
from cwn_amd.synthetic import zinc_like_complexes
dev = torch.device('cuda', 0)
B, S, H, L = 128, int(os.environ.get('S', '4')), 128, 4
pool = [c for i in range(16) for c in zinc_like_complexes(B, 9000 + i, 6)]
packed = PackedComplexes(pool, dev, max_dim=2, with_csr=True)
torch.manual_seed(0)
from cwn_amd.models import EmbedSparseCIN
model = (EmbedSparseCIN if os.environ.get('MODEL') == 'sparse' else EmbedCINpp)(28, 4, 1, L, H, dropout_rate=0.0, max_dim=2, jump_mode=None, nonlinearity='relu', readout='sum', train_eps=False,
                   final_hidden_multiplier=2, final_readout='sum', init_reduce='sum', embed_edge=True, use_coboundaries=True, graph_norm='bn').to(dev).eval()
sb = StaticBatch(packed, B, slots=S, mode='csr')
print('caps', sb.cap_cells, [(k, c) for k, c in zip([(d, key) for d, key, _ in packed._klist], sb._caps)], flush=True)
loader = PackedLoader(packed, batch_size=B, shuffle=True, seed=17)
loader.set_epoch(0)
sb.set_epoch(loader.batches())
g_ = torch.Generator().manual_seed(3)
feats = [[torch.randn(sb.cap_cells[d], H, generator=g_).to(dev) for d in range(3)] for l in range(L)]
MODE = os.environ.get('MODE', 'eager')
def steps(verbose=False):
    sb.fill()
    keep = []
    for slot in sb.slots:
        b = slot.batch
        with slot.dynamic():
            for l, conv in enumerate(model.convs[:int(os.environ.get('NL', '4'))]):
                b.set_xs(feats[l])
                _, outs = conv.propagate_all(*b.get_all_cochain_params(max_dim=2, include_down_features=False))
                if verbose:
                    torch.cuda.synchronize(); print('slot', slot.j, 'layer', l, 'ok', flush=True)
        slot.restore()
        keep.append(outs if int(os.environ.get('NL', '4')) else None)
    return keep
with torch.no_grad():
    if MODE == 'graph':
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            steps(); steps()
        torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode='thread_local'):
            keep = steps()
        run = g.replay
    else:
        run = steps
    for e in range(6):
        loader.set_epoch(1 + e)
        n_rep = sb.set_epoch(loader.batches())
        for r in range(n_rep):
            run()
            torch.cuda.synchronize(); print('epoch', e, 'replay', r, [sb.sizes(j) for j in range(S)], flush=True)
    csr.check_errors(dev)
    print('done'); sys.exit(0)
    for slot in sb.slots:
        b = slot.batch
        with slot.dynamic():
            for l, conv in enumerate(model.convs[:int(os.environ.get('NL', '4'))]):
                b.set_xs(feats[l])
                _, outs = conv.propagate_all(*b.get_all_cochain_params(max_dim=2, include_down_features=False))
                torch.cuda.synchronize(); print('slot', slot.j, 'layer', l, 'ok', flush=True)
        slot.restore()
print('done')
