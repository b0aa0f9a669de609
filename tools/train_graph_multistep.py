"""Is the gap between two graph replays of the training step real?  The same two ZINC-128 steps as ONE graph per step
(cwn_amd.train.TrainStep) and as one graph holding K steps.  usage: train_graph_multistep.py [K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cwn_amd.complex import ComplexBatch
from cwn_amd.models import EmbedSparseCIN
from cwn_amd.synthetic import zinc_like_complexes
from cwn_amd.train import TrainStep, CAPTURE_MODE

dev = torch.device('cuda:0')
torch.manual_seed(0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
model = EmbedSparseCIN(28, 4, 1, 4, 128, dropout_rate=0.0, embed_edge=True, use_coboundaries=True).to(dev)
batches = [ComplexBatch.from_complex_list(zinc_like_complexes(128, i, 6), max_dim=2).to(dev) for i in range(2)]
ts = TrainStep(model, batches, use_graph=True)
for i in range(4):
    ts.step(i % 2)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(200):
    ts.step(i % 2)
torch.cuda.synchronize()
print(f'one graph per step : {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms/step')
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, pool=ts._graphs[0][0][0].pool(), capture_error_mode=CAPTURE_MODE):
    for k in range(K):
        ts._eager(k % 2)
g.replay(); torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(200 // K):
    g.replay()
torch.cuda.synchronize()
print(f'{K} steps per graph  : {(time.perf_counter() - t0) / (200 // K * K) * 1e3:.3f} ms/step')
