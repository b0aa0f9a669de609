"""cProfile of the EAGER training step (host time per launch: ~100 us of Python; the product path replays a graph)."""
import os, sys, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cwn_amd.complex import ComplexBatch
from cwn_amd.models import EmbedSparseCIN
from cwn_amd.synthetic import zinc_like_complexes
from cwn_amd.train import TrainStep
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = EmbedSparseCIN(28, 4, 1, 4, 128, dropout_rate=0.0, embed_edge=True, use_coboundaries=True).to(dev)
batches = [ComplexBatch.from_complex_list(zinc_like_complexes(128, i, 6), max_dim=2).to(dev) for i in range(2)]
ts = TrainStep(model, batches, use_graph=False)
for i in range(4):
    ts.step(i % 2)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(20):
    ts.step(i % 2)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(45)
