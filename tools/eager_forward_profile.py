"""Host time of the eager full forward (EmbedSparseCIN, ZINC-128, inference): total and a cProfile by own time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cwn_amd.models import EmbedSparseCIN
from cwn_amd.complex import ComplexBatch
from cwn_amd.synthetic import zinc_like_complexes
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = EmbedSparseCIN(28, 4, 1, 4, 128, dropout_rate=0.0, embed_edge=True, use_coboundaries=True).to(dev).eval()
b = ComplexBatch.from_complex_list(zinc_like_complexes(128, 0, 6), max_dim=2).to(dev)
x0 = [None if b.cochains[d].x is None else b.cochains[d].x.clone() for d in range(3)]
def fwd():
    for d in range(3):        # (the model overwrites the batch's features layer by layer, as the reference does)
        b.cochains[d]._x = x0[d]
    return model(b)
with torch.no_grad():
    for _ in range(20): fwd()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300): fwd()
    torch.cuda.synchronize()
    print(f'eager model(batch): {(time.perf_counter() - t0) / 300 * 1e6:.0f} us')
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(100): fwd()
    pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(22)
    pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
