"""Who builds CSR plans inside a training step?  Prints a short stack for every csr.build_many call of steady-state steps.
    gpurun -- 'python tools/trace_plan_builds.py'"""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cwn_amd import csr
from cwn_amd.complex import ComplexBatch
from cwn_amd.models import EmbedSparseCIN
from cwn_amd.synthetic import zinc_like_complexes
from cwn_amd.train import TrainStep

dev = torch.device('cuda:0')
torch.manual_seed(0)
model = EmbedSparseCIN(28, 4, 1, 4, 128, dropout_rate=0.0, embed_edge=True, use_coboundaries=True).to(dev)
batches = [ComplexBatch.from_complex_list(zinc_like_complexes(128, i, 6), max_dim=2).to(dev) for i in range(2)]
for b in batches:
    if b.y is None:
        b.y = torch.zeros(b.num_complexes, 1, device=dev)
ts = TrainStep(model, batches, task_type='regression', use_graph=False)
for i in range(4):
    ts.step(i % 2)
ts._skip_upper_plans = os.environ.get('SKIP_UPPER', '1') == '1'
orig = csr.build_many
LOG = []
def traced(adjs, *a, **k):
    adjs = list(adjs)
    st = [f'{os.path.basename(f.filename)}:{f.lineno} {f.name}' for f in traceback.extract_stack()[-9:-1]]
    LOG.append((len(adjs), [(getattr(x, "n_dst", None), getattr(x, "n_entries", None)) for x in adjs][:6], st))
    return orig(adjs, *a, **k)
csr.build_many = traced
import cwn_amd.complex as cc
ts.step(0)
torch.cuda.synchronize()
for n, shapes, st in LOG:
    print(f'build_many of {n} adjacencies {shapes}')
    for s in st:
        print('     ', s)
