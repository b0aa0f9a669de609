"""The device-side item cut of a StaticBatch (cwn_layer_items_build_dev: groups of `group` consecutive complexes per item) swept
over (variant, group): the propagate scope and the full forward of a never-seen epoch, one captured graph each, against the
fixed-batch replay of the host builder's table.
usage: sweep_static_group.py [workload: molhiv | zinc] [batch] ['v:g,v:g,...']"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from cwn_amd.models import EmbedSparseCIN, OGBEmbedSparseCIN
from cwn_amd.packed import PackedComplexes, PackedLoader
from cwn_amd.static_batch import StaticBatch
from cwn_amd.static_graph import StaticForward
from cwn_amd.synthetic import molhiv_like_complexes, zinc_like_complexes

dev = torch.device('cuda', 0)
WL = sys.argv[1] if len(sys.argv) > 1 else 'molhiv'
B = int(sys.argv[2]) if len(sys.argv) > 2 else (512 if WL == 'molhiv' else 128)
COMBOS = [tuple(int(v) for v in c.split(':')) for c in (sys.argv[3] if len(sys.argv) > 3 else '0:4,0:1,0:2,0:3,1:1,1:2,1:3').split(',')]
NB, S, EPOCHS = 32, 16, 4
torch.manual_seed(0)
if WL == 'molhiv':
    model = OGBEmbedSparseCIN(1, 2, 64, dropout_rate=0.0, max_dim=2, readout='mean', final_readout='sum', init_reduce='sum',
                              embed_edge=True, use_coboundaries=True, graph_norm='bn').to(dev).eval()
    gen, H, L = (lambda s: molhiv_like_complexes(B, s, 6)), 64, 2
else:
    model = EmbedSparseCIN(28, 4, 1, 4, 128, dropout_rate=0.0, embed_edge=True, use_coboundaries=True).to(dev).eval()
    gen, H, L = (lambda s: zinc_like_complexes(B, s, 6)), 128, 4
pool = [c for i in range(NB) for c in gen(9000 + i)]
packed = PackedComplexes(pool, dev, max_dim=2, with_csr=True)
loader = PackedLoader(packed, batch_size=B, shuffle=True, seed=17)


def epoch(e):
    loader.set_epoch(e)
    return loader.batches()


def cells(bs):
    return float(sum(int(packed._meta[idx][:, 0:9:3].sum()) for idx in bs)) * L


def run(replay, sb):
    n_rep = sb.set_epoch(epoch(1))
    for _ in range(n_rep):
        replay()
    torch.cuda.synchronize()
    total, t0 = 0.0, time.perf_counter()
    for e in range(EPOCHS):
        bs = epoch(2 + e)
        for _ in range(sb.set_epoch(bs)):
            replay()
        total += cells(bs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return total / dt / 1e6, dt / (EPOCHS * NB) * 1e3


# the fixed-batch replay of the same scope (the host builder's table)
with torch.no_grad():
    fixed = packed.collate(epoch(0)[0])
    g_ = torch.Generator().manual_seed(3)
    ffe = [[torch.randn(fixed.cochains[d].num_cells, H, generator=g_).to(dev) for d in range(3)] for l in range(L)]

    def fixed_prop():
        fixed.block_plan().forget_csr()
        for l, conv in enumerate(model.convs):
            fixed.set_xs(ffe[l])
            _, outs = conv.propagate_all(*fixed.get_all_cochain_params(max_dim=2, include_down_features=False))
        return outs
    for _ in range(3):
        fixed_prop()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        keep = fixed_prop()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        g.replay()
    torch.cuda.synchronize()
    fixed_ms = (time.perf_counter() - t0) / 200 * 1e3
    tab = fixed.block_plan()
    print(f'{WL}-{B}: fixed-batch propagate {fixed_ms:.4f} ms  ({sum(int(fixed.cochains[d].num_cells) for d in range(3)) * L / fixed_ms / 1e3:.0f} M cells/s)', flush=True)

for variant, group in COMBOS:
    try:
        sb = StaticBatch(packed, B, slots=S, variant=variant, group=group)
        sb.reserve_epoch(NB)
        g_ = torch.Generator().manual_seed(3)
        feats = [[torch.randn(sb.cap_cells[d], H, generator=g_).to(dev) for d in range(3)] for l in range(L)]

        def prop_steps():
            sb.fill()
            keep = []
            for slot in sb.slots:
                b, outs = slot.batch, None
                with slot.dynamic():
                    for l, conv in enumerate(model.convs):
                        b.set_xs(feats[l])
                        _, outs = conv.propagate_all(*b.get_all_cochain_params(max_dim=2, include_down_features=False))
                slot.restore()
                keep.append(outs)
            return keep
        with torch.no_grad():
            sb.set_epoch(epoch(0))
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                prop_steps()
                prop_steps()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                keep = prop_steps()
            fit = float(np.mean([bool(v) for e in range(2 + EPOCHS) for v in sb.fits(epoch(e))]))
            p_rate, p_ms = run(gr.replay, sb)
            sf = StaticForward(model, sb)
            f_rate, f_ms = run(sf.replay, sb)
        from cwn_amd import csr
        word = int(csr._err_flag(dev).item())
        print(f'  variant {variant} group {group}: propagate {p_ms:.4f} ms ({p_rate:.0f} M cells/s, {fixed_ms / p_ms:.2f} of fixed)   forward {f_ms:.4f} ms   '
              f'batches that fit {fit:.2f}   error word {word}', flush=True)
        csr._err_flag(dev).zero_()
        del sb, gr, sf, keep, feats
    except Exception as e:
        print(f'  variant {variant} group {group}: {type(e).__name__}: {e}', flush=True)
        torch.cuda.synchronize()
