"""Two ranks sharing ONE GPU over gloo (a control-flow check of TrainStep under data parallelism on a
one-GPU box):  torchrun --nproc-per-node 2 tools/train_2rank_check.py {eager|graph|noreduce} [x = no per-step sync]"""
import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from cwn_amd.models import EmbedSparseCIN
from cwn_amd.synthetic import zinc_like_batch
from cwn_amd.train import TrainStep
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(0)
dist.init_process_group('gloo')
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = EmbedSparseCIN(28, 4, 1, int(os.environ.get("LAYERS", "2")), int(os.environ.get("HIDDEN", "32")), dropout_rate=0.0, embed_edge=True, use_coboundaries=True).to(dev)
batches = [zinc_like_batch(16, seed=100 * rank + i, device=dev) for i in range(2)]
mode = sys.argv[1]
ts = TrainStep(model, batches, use_graph=('eager' not in mode))
if mode == 'noreduce':
    ts.bucket.all_reduce_mean = lambda *a, **k: None
sync = len(sys.argv) < 3
if mode.startswith('noreduce'):
    ts.bucket.all_reduce_mean = lambda *a, **k: None
hist = []
for i in range(40):
    l = ts.step(i % 2)
    if sync:
        torch.cuda.synchronize()
    hist.append(l.clone())
torch.cuda.synchronize()
if rank == 0:
    print(f'[{mode} sync={sync}] hist ' + ' '.join(f'{float(h):.4f}' for h in hist[:12]), flush=True)
print(f'[{mode} sync={sync}] rank {rank} done loss {float(l):.4f}', flush=True)
dist.barrier()
dist.destroy_process_group()
