"""world_size-2 gloo tests (CPU) of the N>1 path: complexes shard disjointly, per-rank batches keep
the reference index layout, and the single flat-bucket gradient all-reduce reproduces the
gradient of the mean loss over both shards."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import cwn_oracle as O
from tests._golden import load, T, dummy_complex as o_complex, state_dict
from tests._product import dummy_complex, list_names


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _loss_and_grads(names, state):
    """Oracle SparseCIN layer (CPU autograd) on the batch of `names`; returns grads per tensor."""
    g = load('sparse_cin_conv.npz')
    cx = O.batch_complexes([o_complex(n) for n in names], max_dim=2)
    gen = torch.Generator().manual_seed(len(names))
    for d in range(len(cx['cochains'])):          # a shard may hold no 2-cells at all (its batch then has two cochains)
        n = cx['cochains'][d]['num_cells']
        cx['cochains'][d]['x'] = torch.randn(n, 8, generator=torch.Generator().manual_seed(100 + d))[:n]
    outs = O.sparse_cin_conv(state, O.all_cochain_params(cx, 2, include_down_features=False), True,
                             training=True)
    return sum(o.pow(2).mean() for o in outs)


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    from cwn_amd.dist import FlatGradBucket, init_from_env, shard, sum_across_ranks
    r, w = init_from_env('gloo')
    assert (r, w) == (rank, world)
    names = list_names('mol')
    mine = shard(names, rank, world)
    # per-rank container batch == oracle batch of the same shard (integer layout)
    from cwn_amd.complex import ComplexBatch
    b = ComplexBatch.from_complex_list([dummy_complex(n) for n in mine], max_dim=2)
    ob = O.batch_complexes([o_complex(n) for n in mine], max_dim=2)
    for d in range(3):
        for k in ('upper_index', 'shared_coboundaries', 'boundary_index', 'batch'):
            a, o = b.cochains[d][k], ob['cochains'][d][k]
            assert (a is None) == (o is None), (rank, d, k)
            assert a is None or torch.equal(a, o), (rank, d, k)
    g = load('sparse_cin_conv.npz')
    state = {k: torch.nn.Parameter(v.clone()) if v.is_floating_point() and 'running' not in k else v
             for k, v in state_dict(g, 'mol_cob_bn/state').items()}
    params = [v for v in state.values() if isinstance(v, torch.nn.Parameter)]
    bucket = FlatGradBucket(params)
    bucket.zero_()
    _loss_and_grads(mine, state).backward()
    local_norm = float(bucket.flat.norm())
    local = bucket.flat.clone()
    bucket.all_reduce_mean()
    plain = bucket.flat.clone()
    # the same gradients as a mean WEIGHTED by the shard sizes (unequal shards): still one collective
    bucket.flat.copy_(local)
    bucket.all_reduce_mean(n_local=len(mine) + rank)      # + rank: make the weights differ
    weighted = bucket.flat.clone()
    bucket.flat.copy_(plain)
    total_cells = sum_across_ranks(float(sum(ob['cochains'][d]['num_cells'] for d in range(3))))
    if rank == 0:
        ret['flat'] = bucket.flat.clone()
        ret['flat_weighted'] = weighted
        ret['cells'] = total_cells
        ret['local_norm'] = local_norm
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_grad_allreduce():
    world = 2
    port = _free_port()
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
        flat = ret['flat']
        flat_w = ret['flat_weighted']
        cells = ret['cells']
    # single-process emulation: mean over the two shards of the per-shard gradient
    from cwn_amd.dist import FlatGradBucket, shard
    g = load('sparse_cin_conv.npz')
    names = list_names('mol')
    ref, ref_w, wsum = None, None, 0.0
    for r in range(world):
        state = {k: torch.nn.Parameter(v.clone()) if v.is_floating_point() and 'running' not in k else v
                 for k, v in state_dict(g, 'mol_cob_bn/state').items()}
        bucket = FlatGradBucket([v for v in state.values() if isinstance(v, torch.nn.Parameter)])
        bucket.zero_()
        _loss_and_grads(shard(names, r, world), state).backward()
        ref = bucket.flat.clone() if ref is None else ref + bucket.flat
        w_r = float(len(shard(names, r, world)) + r)
        ref_w = w_r * bucket.flat if ref_w is None else ref_w + w_r * bucket.flat
        wsum += w_r
    ref /= world
    torch.testing.assert_close(flat, ref, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(flat_w, ref_w / wsum, rtol=1e-5, atol=1e-7)
    total = sum(O.batch_complexes([o_complex(n) for n in names], max_dim=2)['cochains'][d]['num_cells']
                for d in range(3))
    assert cells == total     # the shards cover every cell exactly once


def _worker8(rank, world, port, ret):
    """Config 4 (exp/scripts/cwn-zinc-full.sh:4-34 under DDP) at the world size the node has: rank r takes complexes
    r, r + 8, ... of ONE global batch, differentiates the mean loss of ITS sub-batch, and the flat bucket is
    all-reduced once, weighted by the shard sizes."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    from cwn_amd.dist import FlatGradBucket, init_from_env, shard
    init_from_env('gloo')
    names = list_names('mol')
    mine = shard(names, rank, world)
    g = load('sparse_cin_conv.npz')
    state = {k: torch.nn.Parameter(v.clone()) if v.is_floating_point() and 'running' not in k else v
             for k, v in state_dict(g, 'mol_cob_bn/state').items()}
    bucket = FlatGradBucket([v for v in state.values() if isinstance(v, torch.nn.Parameter)])
    bucket.zero_()
    _loss_and_grads(mine, state).backward()
    bucket.all_reduce_mean(n_local=len(mine))
    if rank in (0, world - 1):
        ret[f'flat{rank}'] = bucket.flat.clone()
        ret[f'n{rank}'] = len(mine)
    dist.barrier()
    dist.destroy_process_group()


def test_eight_rank_gradient_is_the_weighted_mean_of_eight_sub_batches():
    """SURVEY.md 7.2 / VERDICT r2 item 6: config-4 parity is defined against a CPU emulation that averages the
    gradients of EIGHT sub-batches (BatchNorm statistics are per shard under DDP), not against the single-batch run.
    Eight gloo ranks, 19 complexes (shards of 3, 3, 3, 2, ...: unequal), one collective: the reduced bucket equals the
    shard-size-weighted mean of the eight per-shard oracle gradients, and is the same on the first and the last rank."""
    world, port = 8, _free_port()
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_worker8, args=(world, port, ret), nprocs=world, join=True)
        flat0, flat7, n0, n7 = ret['flat0'], ret[f'flat{world - 1}'], ret['n0'], ret[f'n{world - 1}']
    from cwn_amd.dist import FlatGradBucket, shard
    g = load('sparse_cin_conv.npz')
    names = list_names('mol')
    sizes = [len(shard(names, r, world)) for r in range(world)]
    assert sum(sizes) == len(names) and (n0, n7) == (sizes[0], sizes[-1]) and len(set(sizes)) > 1
    ref = None
    for r in range(world):
        state = {k: torch.nn.Parameter(v.clone()) if v.is_floating_point() and 'running' not in k else v
                 for k, v in state_dict(g, 'mol_cob_bn/state').items()}
        bucket = FlatGradBucket([v for v in state.values() if isinstance(v, torch.nn.Parameter)])
        bucket.zero_()
        _loss_and_grads(shard(names, r, world), state).backward()
        ref = sizes[r] * bucket.flat if ref is None else ref + sizes[r] * bucket.flat
    ref = ref / sum(sizes)
    torch.testing.assert_close(flat0, ref, rtol=1e-5, atol=1e-7)
    assert torch.equal(flat0, flat7)                      # every rank holds the same reduced gradient


def test_shard_is_a_partition():
    from cwn_amd.dist import shard
    items = list(range(23))
    for world in (1, 2, 4, 8):
        parts = [shard(items, r, world) for r in range(world)]
        assert sorted(x for p in parts for x in p) == items
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_bucket_views_are_the_grads():
    from cwn_amd.dist import FlatGradBucket
    lin = torch.nn.Linear(3, 2)
    bucket = FlatGradBucket(lin.parameters())
    lin(torch.ones(4, 3)).sum().backward()
    assert bucket.flat.numel() == 12 and bucket.flat.abs().sum() > 0
    assert lin.weight.grad.data_ptr() == bucket.flat.data_ptr()
    assert bucket.all_reduce_mean() is None      # no process group: no-op


class _Layered(torch.nn.Module):
    """A layered toy network with the attribute the trainer cuts at (`convs`), an embedding in front, a head
    behind and a parameter the loss never reaches."""

    def __init__(self):
        super().__init__()
        self.head = torch.nn.Linear(4, 1)
        self.convs = torch.nn.ModuleList(torch.nn.Linear(4, 4) for _ in range(4))
        self.embed = torch.nn.Embedding(5, 4)
        self.unused = torch.nn.Linear(2, 2)

    def forward(self, idx, skip=False, jk=False):
        x = x0 = self.embed(idx)
        js = []
        for c in self.convs:
            x = c(x).relu()
            js.append(x)
        y = self.head(torch.stack(js, -1).max(-1)[0] if jk else x).pow(2).sum()
        return y + (x0.sum() if skip else 0)


def _staged_setup(seed=0):
    from cwn_amd.dist import FlatGradBucket, StagedBackward
    torch.manual_seed(seed)
    net = _Layered()
    st = StagedBackward([net.convs[0], net.convs[2]])
    st.begin()
    so = st.stages(net(torch.tensor([0, 1, 1, 4])), list(net.parameters()))
    bucket = FlatGradBucket(net.parameters(), so, st.n_stages)
    sp = [[p for p in bucket.params if so[id(p)] == st.n_stages - 1 - j] for j in range(st.n_stages)]
    return net, st, so, bucket, sp


def test_staged_backward_equals_the_monolithic_backward():
    """The backward in three pieces cut behind layers 2 and 0: stages read off the autograd graph, every chunk of
    the bucket untouched until its piece has run, the result bit-identical to loss.backward(); networks whose
    layers cannot be cut (a skip connection around the layers, jumping knowledge) are recognised."""
    net, st, so, bucket, sp = _staged_setup()
    stage = {n: so[id(p)] for n, p in net.named_parameters()}
    assert stage['head.weight'] == stage['convs.3.bias'] == 2 and stage['convs.2.weight'] == stage['convs.1.weight'] == 1
    assert stage['convs.0.weight'] == stage['embed.weight'] == stage['unused.weight'] == 0
    assert [hi - lo for lo, hi in bucket.chunks] == [28, 40, 48] and bucket.chunks[0][0] == 0
    idx = torch.tensor([0, 3, 3, 2])
    bucket.zero_()
    st.begin()
    loss = net(idx)
    for j in range(st.n_stages):
        st.piece(j, loss, sp[j])
        assert bucket.chunk(j).any()
        assert not any(bucket.chunk(c).any() for c in range(j + 1, st.n_stages)), j
    got = bucket.flat.clone()
    bucket.zero_()
    net(idx).backward()
    assert torch.equal(got, bucket.flat)
    for kw in (dict(skip=True), dict(jk=True)):
        st.begin()
        assert st.stages(net(idx, **kw), list(net.parameters())) is None, kw
    with pytest.raises(ValueError):
        from cwn_amd.dist import FlatGradBucket
        FlatGradBucket(net.parameters(), {id(net.head.weight): 3}, 3)


def _staged_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from cwn_amd.dist import init_from_env
    init_from_env('gloo')
    net, st, so, bucket, sp = _staged_setup()
    idx = torch.tensor([[0, 1, 1, 4], [2, 2, 3, 0]][rank])
    n_local = 3 + 2 * rank
    bucket.zero_()
    st.begin()
    loss = net(idx)
    for j in range(st.n_stages):
        st.piece(j, loss, sp[j])
        bucket.reduce_chunk(j, n_local)          # in flight while the next piece runs
    bucket.finish()
    if rank == 0:
        ret['flat'] = bucket.flat.clone()
        ret['order'] = [so[id(p)] for p in bucket.params]
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_chunked_reduce_inside_the_backward():
    """reduce_chunk after every piece of the staged backward == the weighted mean of the per-rank gradients
    (what all_reduce_mean gives after a monolithic backward), count element riding with the last chunk."""
    world, port = 2, _free_port()
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_staged_worker, args=(world, port, ret), nprocs=world, join=True)
        flat, order = ret['flat'], ret['order']
    assert order == sorted(order, reverse=True)
    ref, wsum = None, 0.0
    for r in range(world):
        net, st, so, bucket, sp = _staged_setup()
        bucket.zero_()
        net(torch.tensor([[0, 1, 1, 4], [2, 2, 3, 0]][r])).backward()
        w = 3.0 + 2 * r
        ref = w * bucket.flat if ref is None else ref + w * bucket.flat
        wsum += w
    torch.testing.assert_close(flat, ref / wsum, rtol=1e-6, atol=1e-7)


def _tensor_count_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from cwn_amd.dist import init_from_env
    init_from_env('gloo')
    out = {}
    for name, counts in (('mixed', [5, 0]), ('none', [0, 0])):
        net, st, so, bucket, sp = _staged_setup()
        bucket.zero_()
        net(torch.tensor([[0, 1, 1, 4], [2, 2, 3, 0]][rank])).backward()
        n_local = torch.tensor([counts[rank]], dtype=torch.int64)            # a static batch: the count is a (device) tensor
        bucket.all_reduce_mean(n_local=n_local)
        out[name] = (bucket.flat.clone(), float(bucket.global_count()))
        # ... and the chunked form of the staged backward
        net, st, so, bucket, sp = _staged_setup()
        bucket.zero_()
        st.begin()
        loss = net(torch.tensor([[0, 1, 1, 4], [2, 2, 3, 0]][rank]))
        for j in range(st.n_stages):
            st.piece(j, loss, sp[j])
            bucket.reduce_chunk(j, n_local)
        bucket.finish()
        out[name + '_chunked'] = (bucket.flat.clone(), float(bucket.global_count()))
    ret[rank] = out
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_reduce_with_device_side_sample_counts():
    """The sample count of a static batch is a tensor (cwn_amd/static_graph.py: no host sync): all_reduce_mean / reduce_chunk
    take it as the rank's weight.  A rank WITHOUT samples (the empty tail of its epoch) contributes nothing and receives the
    other's gradient; a step in which no rank holds a sample leaves zeros (and a global count of 0: the optimizer skips it)."""
    world, port = 2, _free_port()
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_tensor_count_worker, args=(world, port, ret), nprocs=world, join=True)
        got = {r: ret[r] for r in range(world)}
    net, st, so, bucket, sp = _staged_setup()
    bucket.zero_()
    net(torch.tensor([0, 1, 1, 4])).backward()                                # rank 0's own gradient
    want = bucket.flat.clone()
    for r in range(world):
        for name in ('mixed', 'mixed_chunked'):
            flat, cnt = got[r][name]
            assert cnt == 5.0
            torch.testing.assert_close(flat, want, rtol=1e-6, atol=1e-7, msg=f'rank {r} {name}')
        for name in ('none', 'none_chunked'):
            flat, cnt = got[r][name]
            assert cnt == 0.0 and float(flat.abs().max()) == 0.0 and bool(torch.isfinite(flat).all()), (r, name)


def test_packed_loader_shards_every_global_batch_across_the_ranks():
    """cwn_amd.packed.PackedLoader (the DataLoader of data/data_loading.py:84-111 over the packed dataset): the ranks
    split each global batch, see the same number of batches, cover a split exactly once per epoch, and reshuffle
    identically (seed + epoch) -- index logic only, the collate itself is a GPU test."""
    import numpy as np
    from cwn_amd.packed import PackedLoader

    class Fake:
        num = 103

        def collate(self, idx):
            return [int(i) for i in idx]
    split = np.arange(3, 103, dtype=np.int64)[::-1].copy()                 # a 'train' split of 100 ids
    for world in (1, 2, 4):
        for bs, shuffle, drop in ((10, False, False), (16, True, True), (16, True, False), (7, False, False)):
            loaders = [PackedLoader(Fake(), bs, shuffle, indices=split, drop_last=drop, seed=5, rank=r, world=world)
                       for r in range(world)]
            epochs = []
            for _ in range(2):
                per = [list(l) for l in loaders]
                assert len({len(p) for p in per}) == 1 and all(len(p) == len(l) for p, l in zip(per, loaders))
                # batch k of all ranks together = global batch k: disjoint, sizes within one of each other
                for k in range(len(per[0])):
                    parts = [p[k] for p in per]
                    assert max(map(len, parts)) - min(map(len, parts)) <= 1 and sum(map(len, parts)) <= bs
                seen = sorted(i for p in per for b in p for i in b)
                assert len(seen) == len(set(seen)) and set(seen) <= set(split.tolist())
                full = (100 // bs) * bs
                assert len(seen) == (full if drop or (100 - full) < world else 100)
                epochs.append(per)
            assert (epochs[0] != epochs[1]) == shuffle                      # a new permutation per epoch
            if not shuffle and world == 1:
                assert epochs[0][0][0] == split[:bs].tolist()
    with pytest.raises(IndexError):
        PackedLoader(Fake(), 4, indices=[0, 103])
    with pytest.raises(ValueError):
        PackedLoader(Fake(), 4, rank=2, world=2)
    assert len(PackedLoader(Fake(), 3, world=4)) == 0                       # a batch smaller than the world: nothing to shard


# ---- world 8: the staged (chunk-by-chunk) reduce and the one-collective fallback of a jumping-knowledge model -----------------
def _idx_of(rank):
    return torch.tensor([(rank + k * (rank % 3 + 1)) % 5 for k in range(4)])


def _staged_worker8(rank, world, port, ret, jk):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from cwn_amd.dist import FlatGradBucket, init_from_env
    init_from_env('gloo')
    n_local = 1 + (3 * rank) % 5                                     # unequal shards: the mean is weighted
    if jk:
        # the loss reads every layer's output (jump_mode: the REDDIT config): the backward cannot be cut -- recognised on the
        # autograd graph -- and the step keeps ONE collective behind the whole backward
        net, st, so, bucket, sp = _staged_setup()
        st.begin()
        assert st.stages(net(_idx_of(rank), jk=True), list(net.parameters())) is None
        bucket = FlatGradBucket(net.parameters())
        bucket.zero_()
        net(_idx_of(rank), jk=True).backward()
        bucket.all_reduce_mean(n_local=n_local)
    else:
        net, st, so, bucket, sp = _staged_setup()
        bucket.zero_()
        st.begin()
        loss = net(_idx_of(rank))
        for j in range(st.n_stages):
            st.piece(j, loss, sp[j])
            bucket.reduce_chunk(j, n_local)                          # in flight while the next piece runs
        bucket.finish()
    if rank in (0, world - 1):
        ret[f'flat{rank}'] = bucket.flat.clone()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('jk', [False, True])
def test_eight_rank_staged_reduce_and_jumping_knowledge_fallback(jk):
    """BASELINE configs[3] at the world size the node has (exp/scripts/cwn-zinc-full.sh:4-34 under DDP), without hardware:
    eight gloo ranks with unequal shards run (a) the staged backward with the chunk-by-chunk reduce issued inside it and
    (b) the one-collective step a jumping-knowledge model falls back to; first and last rank must hold the shard-weighted
    mean of the eight per-rank gradients."""
    from cwn_amd.dist import FlatGradBucket
    world, port = 8, _free_port()
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_staged_worker8, args=(world, port, ret, jk), nprocs=world, join=True)
        first, last = ret['flat0'], ret[f'flat{world - 1}']
    assert torch.equal(first, last)
    ref, wsum = None, 0.0
    for r in range(world):
        net, st, so, bucket, sp = _staged_setup()
        if jk:
            bucket = FlatGradBucket(net.parameters())
        bucket.zero_()
        net(_idx_of(r), jk=jk).backward()
        w = float(1 + (3 * r) % 5)
        ref = w * bucket.flat if ref is None else ref + w * bucket.flat
        wsum += w
    torch.testing.assert_close(first, ref / wsum, rtol=1e-6, atol=1e-7)
