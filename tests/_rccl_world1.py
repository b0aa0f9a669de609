"""RCCL on the one GPU a test box has: a process group of ONE rank over the "nccl" backend (= RCCL on
ROCm) with the data-parallel form of the training step forced (cwn_amd.dist.FORCE_DP): the backward cut into chunks, each
chunk's all-reduce issued to RCCL between the captured pieces -- the buffer reduced with itself --, the optimizer graph
behind the last one.  Every step must equal the world-1 single-graph step; what stays unknown after this is the xGMI wire.

Run as a script by tests/test_gpu_rccl.py (its own process: a process group is process-wide state), prints one JSON line."""
import json
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def main():
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{_free_port()}', rank=0, world_size=1, device_id=dev)
    out = {'backend': dist.get_backend(), 'world': dist.get_world_size()}

    # count what reaches the backend
    calls = {'all_reduce': 0, 'bytes': 0}
    real = dist.all_reduce

    def counted(t, *a, **k):
        calls['all_reduce'] += 1
        calls['bytes'] += t.numel() * t.element_size()
        return real(t, *a, **k)
    dist.all_reduce = counted

    from cwn_amd import dist as cd
    from cwn_amd.complex import ComplexBatch
    from cwn_amd.models import EmbedSparseCIN, SparseCIN
    from cwn_amd.packed import PackedComplexes
    from cwn_amd.static_batch import StaticBatch
    from cwn_amd.static_graph import StaticTrainStep
    from cwn_amd.synthetic import zinc_like_complexes
    from cwn_amd.train import TrainStep

    def zinc_model(hidden=64, layers=4, seed=0, jump=None):
        torch.manual_seed(seed)
        return EmbedSparseCIN(28, 4, 1, layers, hidden, dropout_rate=0.0, max_dim=2, jump_mode=jump, nonlinearity='relu',
                              readout='sum', train_eps=False, final_hidden_multiplier=2, final_readout='sum', init_reduce='sum',
                              embed_edge=True, use_coboundaries=True, graph_norm='bn').to(dev)

    def batches(n=3, B=32):
        bs = []
        for i in range(n):
            b = ComplexBatch.from_complex_list(zinc_like_complexes(B, seed=50 + i, max_ring=6), max_dim=2)
            bs.append(b.to(dev))
        return bs

    def compare(name, make, bs_a, bs_b, steps=4, **kw):
        """the plain world-1 step (one graph) against the forced data-parallel form over RCCL, from the same state"""
        ma, mb = make(), make()
        mb.load_state_dict(ma.state_dict())
        cd.FORCE_DP = False
        plain = TrainStep(ma, bs_a, lr=1e-3, use_graph=True)
        assert plain.world == 1 and plain.n_stages == 1
        cd.FORCE_DP = True
        before = calls['all_reduce']
        forced = TrainStep(mb, bs_b, lr=1e-3, use_graph=True, **kw)
        assert forced.world == 2
        la, lb = [], []
        for i in range(steps):
            la.append(float(plain.step(i % len(bs_a))))
            lb.append(float(forced.step(i % len(bs_b))))
        torch.cuda.synchronize()
        cd.FORCE_DP = False
        worst = max(float((a.detach() - b.detach()).abs().max()) for a, b in zip(ma.parameters(), mb.parameters()))
        out[name] = {'pieces': forced.n_stages, 'loss_plain': la, 'loss_forced': lb, 'param_delta': worst,
                     'all_reduce_calls': calls['all_reduce'] - before, 'bucket_bytes': int(forced.bucket.flat.numel() * 4)}
        # step 0 from identical states: the loss is computed before any collective -> equal to rounding; afterwards one Adam
        # sign flip of a noise-level gradient per step at most (lr = 1e-3)
        assert abs(la[0] - lb[0]) <= 1e-5 * max(1.0, abs(la[0])), (name, la, lb)
        for a, b in zip(la[1:], lb[1:]):
            assert abs(a - b) <= 1e-2 * max(1.0, abs(a)), (name, la, lb)
        assert worst <= 2 * 1e-3 * steps * 1.1, (name, worst)
        return forced

    # (1) the staged form: 4 layers -> the backward in pieces, one asynchronous all-reduce per piece on RCCL's stream
    f = compare('staged', lambda: zinc_model(64, 4, seed=1), batches(), batches())
    assert f.n_stages > 1 and out['staged']['all_reduce_calls'] >= 4 * f.n_stages
    # (2) a jumping-knowledge model cannot be cut: ONE collective behind the backward
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        f = compare('jumping_knowledge', lambda: zinc_model(64, 3, seed=2, jump='cat'), batches(), batches())
    assert f.n_stages == 1 and out['jumping_knowledge']['all_reduce_calls'] >= 4
    # (3) gradient values through the collective: reduce(g) with itself, weighted mean = g  (one eager step, no graph)
    cd.FORCE_DP = True
    m = zinc_model(64, 2, seed=3)
    bs = batches(1)
    t = TrainStep(m, bs, lr=1e-3, use_graph=False, stages=1)
    t._forward_backward(0)
    g0 = t.bucket.flat.clone()
    t.bucket.all_reduce_mean(n_local=bs[0].num_complexes)
    torch.cuda.synchronize()
    rel = float((t.bucket.flat - g0).abs().max()) / max(1e-30, float(g0.abs().max()))
    out['self_reduce_rel_err'] = rel
    assert rel <= 1e-6 and float(t.bucket.global_count()) == bs[0].num_complexes
    cd.FORCE_DP = False

    # (4) static batches (never-seen batches, device-side counts) in the data-parallel form, incl. an empty tail slot
    pool = zinc_like_complexes(150, seed=3, max_ring=6, n_lo=9, n_hi=28)
    p = PackedComplexes(pool, dev, max_dim=2, with_csr=True)
    rng = np.random.default_rng(0)
    perm = rng.permutation(len(pool))
    B = 40
    ep = [perm[0:B], perm[B:2 * B], perm[2 * B:2 * B + 25]]
    ma, mb = zinc_model(64, 2, seed=4), zinc_model(64, 2, seed=4)
    mb.load_state_dict(ma.state_dict())
    sa, sb = StaticBatch(p, B, slots=2), StaticBatch(p, B, slots=2)
    cd.FORCE_DP = False
    one = StaticTrainStep(ma, sa, lr=1e-3)
    cd.FORCE_DP = True
    before = calls['all_reduce']
    two = StaticTrainStep(mb, sb, lr=1e-3)
    assert two.world == 2
    sa.set_epoch(ep)
    sb.set_epoch(ep)
    la, lb = [], []
    for r in range(2):
        la += [float(x) for x in one.step()]
        lb += [float(x) for x in two.step()]
    torch.cuda.synchronize()
    cd.FORCE_DP = False
    out['static'] = {'loss_plain': la, 'loss_forced': lb, 't_plain': int(one.opt.t), 't_forced': int(two.opt.t),
                     'all_reduce_calls': calls['all_reduce'] - before}
    assert int(one.opt.t) == int(two.opt.t) == 3                      # the empty fourth slot took no step on either path
    assert abs(la[0] - lb[0]) <= 1e-5 * max(1.0, abs(la[0]))
    for a, b in zip(la[1:3], lb[1:3]):
        assert abs(a - b) <= 1e-2 * max(1.0, abs(a)), (la, lb)
    assert la[3] != la[3] and lb[3] != lb[3]
    out['all_reduce_calls'] = calls['all_reduce']
    out['all_reduce_bytes'] = calls['bytes']
    dist.all_reduce = real
    dist.barrier()
    dist.destroy_process_group()
    print('RCCL_WORLD1 ' + json.dumps(out))


if __name__ == '__main__':
    main()
