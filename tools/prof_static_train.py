"""One fixed-batch TrainStep and one csr-mode StaticTrainStep of the REDDIT-like configuration, a few steps each (run under
rocprofv3 --kernel-trace --stats: tools/prof_static_train.sh) -- which launches does the static step add?
usage: prof_static_train.py fixed|static"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cwn_amd.complex import ComplexBatch
from cwn_amd.models import SparseCIN
from cwn_amd.packed import PackedComplexes
from cwn_amd.static_batch import StaticBatch
from cwn_amd.static_graph import StaticTrainStep
from cwn_amd.synthetic import reddit_like_complexes
from cwn_amd.train import TrainStep
dev = torch.device('cuda', 0)
which = sys.argv[1]
torch.manual_seed(0)
model = SparseCIN(1, 2, 4, 64, dropout_rate=0.0, max_dim=2, jump_mode='cat', readout='sum', use_coboundaries=False, graph_norm='id').to(dev)
with torch.no_grad():
    for p_ in model.parameters():
        p_.mul_(0.3)
B, S = 32, 4
pool = [c for i in range(4) for c in reddit_like_complexes(B, 50 + i)]
if which == 'fixed':
    bs = [ComplexBatch.from_complex_list(pool[i * B:(i + 1) * B], max_dim=2).to(dev) for i in range(4)]      # (the static run's four batches)
    for b in bs:
        b.y = torch.zeros(b.num_complexes, dtype=torch.long, device=dev)
    ts = TrainStep(model, bs, task_type='classification', use_graph=True)
    for i in range(24):
        ts.step(i % 4)
else:
    for c in pool:
        c.y = torch.zeros(1, dtype=torch.long)
    packed = PackedComplexes(pool, dev, max_dim=2, with_csr=True)
    caps = None
    if os.environ.get('TIGHT_CAPS') == '1':      # (experiment: capacities = the largest of these four batches, not the 6-sigma bound)
        m = packed._meta
        D = packed.max_dim + 1
        per = [m[k * B:(k + 1) * B] for k in range(4)]
        caps = {'cells': [int(max(int(x[:, 3 * d].sum()) for x in per)) + 8 + d for d in range(D)]}
        for k, (d, key, pk) in enumerate(packed._klist):
            if key in ('upper_index', 'lower_index', 'boundary_index'):
                caps[(d, key)] = int(max(int(x[:, 3 * D + k].sum()) for x in per)) + 16 + k
    sb = StaticBatch(packed, B, slots=S, mode='csr', caps=caps)
    print('capacities', sb.cap_cells)
    ts = StaticTrainStep(model, sb, task_type='classification', lr=1e-3)
    perm = np.arange(len(pool))
    for e in range(6):
        ts.run_epoch([perm[k * B:(k + 1) * B] for k in range(4)], keep_losses=False)
torch.cuda.synchronize()
print('done', which)
