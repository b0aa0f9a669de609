"""Two ranks sharing ONE GPU over gloo (a control-flow check of TrainStep under data parallelism on a
one-GPU box):  torchrun --nproc-per-node 2 tools/train_2rank_check.py {eager|graph|noreduce} [x = no per-step sync]
               torchrun --nproc-per-node 2 tools/train_2rank_check.py compare
`compare`: the step with the backward cut into pieces and the all-reduce issued chunk by chunk inside it (the
data-parallel default) against the step with ONE all-reduce after the backward, same model, same shards, both as
hipGraphs and eagerly; prints `compare OK` per rank (tests/test_gpu_parity.py runs it)."""
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
batches = [zinc_like_batch(16 + 4 * rank, seed=100 * rank + i, device=dev) for i in range(2)]    # unequal shards
mode = sys.argv[1]
if mode == 'compare':
    import copy

    def reduced_gradient(ts, i):
        """forward + backward + the step's collective(s), no optimiser step; gradients in model order"""
        n_local = ts.batches[i].num_complexes
        if ts.staged is None:
            ts._forward_backward(i)
            ts.bucket.all_reduce_mean(n_local=n_local)
        else:
            for j in range(ts.n_stages):
                ts._forward_backward(i, [j])
                ts.bucket.reduce_chunk(j, n_local)
            ts.bucket.finish()
        return torch.cat([p.grad.flatten() for p in ts.model.parameters()])

    worst = 0.0
    for use_graph in (True, False):
        m1, m2 = copy.deepcopy(model), copy.deepcopy(model)
        one = TrainStep(m1, batches, use_graph=use_graph, stages=1)
        cut = TrainStep(m2, batches, use_graph=use_graph)
        assert one.n_stages == 1 and cut.n_stages == min(4, len(model.convs)) > 1, (one.n_stages, cut.n_stages)
        # the reduced gradient itself: chunk by chunk inside the backward == one all-reduce behind it
        for i in range(2):
            ga, gb = reduced_gradient(one, i), reduced_gradient(cut, i)
            rel = float((ga - gb).norm() / ga.norm())
            top = float((ga - gb).abs().max() / ga.abs().max())
            assert rel < 1e-5 and top < 1e-5, (use_graph, i, rel, top)
            worst = max(worst, rel)
        # the steps (hipGraphs / eager): losses agree up to what Adam makes of summation noise (a noise-level
        # gradient moves its weight by +-lr whatever its size), and both ranks end with the same model, bit for bit
        for i in range(6):
            la, lb = one.step(i % 2), cut.step(i % 2)
            assert torch.isfinite(lb) and abs(float(la) - float(lb)) < 3e-2 * abs(float(la)), (use_graph, i, float(la), float(lb))
        torch.cuda.synchronize()
        for m in (m1, m2):
            flat = torch.cat([p.detach().flatten() for p in m.parameters()])
            other = flat.clone()
            dist.broadcast(other, src=0)
            assert torch.equal(flat, other), 'ranks diverged'
    print(f'compare OK rank {rank} worst relative distance of the reduced gradient {worst:.3e}', flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0)
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
