import os, sys
sys.path.insert(0, '/root/repo')
import torch
from torch.profiler import profile, ProfilerActivity
from cwn_amd.complex import ComplexBatch
from cwn_amd.models import EmbedSparseCIN
from cwn_amd.synthetic import zinc_like_complexes
from cwn_amd.train import TrainStep
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = EmbedSparseCIN(28, 4, 1, 4, 128, dropout_rate=0.0, embed_edge=True, use_coboundaries=True).to(dev)
batches = [ComplexBatch.from_complex_list(zinc_like_complexes(128, i, 6), max_dim=2).to(dev) for i in range(2)]
ts = TrainStep(model, batches, use_graph=False)
for i in range(3): ts.step(i % 2)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    ts.step(0)
torch.cuda.synchronize()
import collections
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ('aten::add', 'aten::add_', 'aten::fill_', 'aten::zero_', 'aten::copy_', 'aten::mul', 'aten::zeros', 'aten::clone', 'aten::contiguous', 'aten::div', 'aten::gt', 'aten::sum'):
        st = [s for s in (ev.stack or []) if '/root/repo' in s]
        key = (ev.name, st[0].split('/root/repo/')[-1] if st else '(autograd engine / torch)')
        cnt[key] += 1
for (name, where), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:40]:
    print(f'{c:4d} {name:16s} {where}')
