"""Which framework (at::native) kernels does one eager training step launch, and from where?  One step of
cwn_amd.train.TrainStep (ZINC-128) under torch.profiler with Python stacks; every aten op that launched a kernel is printed
with the innermost cwn_amd frame of its stack ('autograd engine' = an accumulation the engine itself inserted).
usage: trace_native_kernels.py [batch]"""
import os
import sys
from collections import Counter
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from cwn_amd.complex import ComplexBatch
from cwn_amd.models import EmbedSparseCIN
from cwn_amd.synthetic import zinc_like_complexes
from cwn_amd.train import TrainStep

dev = torch.device('cuda:0')
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
model = EmbedSparseCIN(28, 4, 1, 4, 128, dropout_rate=0.0, embed_edge=True, use_coboundaries=True).to(dev)
batches = [ComplexBatch.from_complex_list(zinc_like_complexes(B, i, 6), max_dim=2).to(dev) for i in range(2)]
ts = TrainStep(model, batches, use_graph=False)
for i in range(3):
    ts.step(i % 2)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    ts.step(1)
    torch.cuda.synchronize()
rows = Counter()
n_kernels = 0
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
        continue
    if not e.name.startswith('aten::'):
        continue
    # only leaves: an op whose child also launched the kernel would be counted twice
    if any(c.kernels for c in e.cpu_children):
        continue
    where = 'autograd engine / no python frame'
    for fr in e.stack or []:
        if 'cwn_amd' in fr or 'tools/' in fr:
            where = fr.strip()
            break
    else:
        # ops issued inside an autograd Function's backward run on the engine's thread: walk up the profiler's own tree
        p = e.cpu_parent
        chain = []
        while p is not None:
            chain.append(p.name)
            p = p.cpu_parent
        named = [c for c in chain if not c.startswith('aten::')]
        if named:
            where = ' < '.join(named[:3])
    rows[(e.name, where + '  ' + str([tuple(x) for x in (e.input_shapes or []) if x][:2]))] += len(e.kernels)
    n_kernels += len(e.kernels)
print(f'{n_kernels} framework kernels in one step:')
for (name, where), n in sorted(rows.items(), key=lambda kv: -kv[1]):
    print(f'{n:4d}  {name:28s} {where}')
