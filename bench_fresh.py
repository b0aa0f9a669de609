"""bench_fresh.py -- the never-seen-batch legs of bench.py (`secondary.fresh_batches`): the propagate scope, the full forward and
the full training step over >= 64 distinct shuffled batches per epoch drawn by cwn_amd.packed.PackedLoader, each scope ONE
captured graph over a StaticBatch (device-side collate + tables + item tables or CSR plans inside the graph), timed over at
least CWN_BENCH_FRESH_MIN_MS of epochs.  Imported by bench.py; tools/prof_fresh*.sh run its legs under rocprofv3."""
import os
import sys
import time

import torch

CAPTURE_MODE = 'thread_local'


def fresh_batches_leg(args, model, gen, dev, H, L, rank, fixed_cells_per_s, fixed_forward_ms, fixed_train_ms, task='regression', mode='blocked'):
    """secondary.fresh_batches: propagate scope, full forward and full training step over >= 64 distinct shuffled batches per
    epoch drawn by cwn_amd.packed.PackedLoader, each scope ONE captured graph over a StaticBatch (see the call site)."""
    import copy
    import numpy as np
    from cwn_amd import csr
    from cwn_amd.packed import PackedComplexes, PackedLoader
    from cwn_amd.static_batch import StaticBatch
    from cwn_amd.static_graph import StaticTrainStep
    # (mode 'csr' -- REDDIT-like hub complexes, CIN++ layers: large batches, a smaller pool and fewer steps per replay)
    # (distinct batches per epoch: 64 at the headline's batch size; the pool is generated in pure Python -- ~1 ms a molecule --
    #  so the batch-512 workloads of the default run draw 24: CWN_BENCH_FRESH_BATCHES=64 for the long form)
    NB = int(os.environ.get('CWN_BENCH_FRESH_BATCHES', ('64' if args.batch <= 128 else '24') if mode == 'blocked' else '16'))
    S = int(os.environ.get('CWN_BENCH_FRESH_SLOTS', '16' if mode == 'blocked' else '8'))      # (8: 634 M cells/s on the propagate scope, 16: 659 M, 32: 668 M)
    EPOCHS = int(os.environ.get('CWN_BENCH_FRESH_EPOCHS', '6'))
    B = args.batch
    pool = [c for i in range(NB) for c in gen(9000 + 1000 * rank + i)]
    packed = PackedComplexes(pool, dev, max_dim=2, with_csr=True)
    loader = PackedLoader(packed, batch_size=B, shuffle=True, seed=17)

    def epoch(e):
        loader.set_epoch(e)
        return loader.batches()

    def cells(bs):
        return float(sum(int(packed._meta[idx][:, 0:9:3].sum()) for idx in bs)) * L

    sb = StaticBatch(packed, B, slots=S, mode=mode)
    sb.reserve_epoch(NB)
    model = model.eval()
    g_ = torch.Generator().manual_seed(3)
    # (a layer's input width: the first layer of a model without an embedding front takes the dataset's own features)
    w_in = [int(getattr(conv.mp_levels[0].update_up_nn[0], 'in_features', H)) for conv in model.convs]
    feats = [[torch.randn(sb.cap_cells[d], w_in[l], generator=g_).to(dev) for d in range(3)] for l in range(L)]

    def prop_steps(n_slots=None):
        sb.fill(n_slots)
        keep = []
        for slot in (sb.slots if n_slots is None else sb.slots[:n_slots]):
            b, outs = slot.batch, None
            with slot.dynamic():
                for l, conv in enumerate(model.convs):
                    b.set_xs(feats[l])
                    _, outs = conv.propagate_all(*b.get_all_cochain_params(max_dim=2, include_down_features=False))
            slot.restore()
            keep.append(outs)
        return keep

    def graph_of(fn):
        with torch.no_grad():
            sb.set_epoch(epoch(0))
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fn()
                fn()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
                keep = fn()
        return g, keep

    MIN_MS = float(os.environ.get('CWN_BENCH_FRESH_MIN_MS', '40'))
    epochs_used = [EPOCHS]

    def time_epochs(one):
        """`one(batches)` runs an epoch of NB never-seen batches (the per-epoch permutation upload included).  Timed: at least
        EPOCHS epochs and at least MIN_MS of them (an epoch of the propagate scope lasts ~1 ms: the host's start-up before the
        first replay -- drawing the permutation, the upload -- would be 5 % of a 6-epoch region); every epoch its own
        permutation.  -> cells / s, ms per step, epochs"""
        one(epoch(1))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        one(epoch(1))
        torch.cuda.synchronize()
        n = int(min(96, max(EPOCHS, -(-MIN_MS * 1e-3 // max(time.perf_counter() - t0, 1e-5)))))
        epochs_used[0] = max(epochs_used[0], n)
        t0, seen = time.perf_counter(), []
        for e in range(n):
            bs_ = epoch(2 + e)
            one(bs_)
            seen.append(bs_)
        torch.cuda.synchronize()
        dt_ = time.perf_counter() - t0
        total = sum(cells(bs_) for bs_ in seen)          # (the bench's own bookkeeping: after the clock stops)
        return total / dt_, dt_ / (n * NB) * 1e3, n

    out = {'distinct_batches_per_epoch': NB, 'epochs_timed_at_least': EPOCHS, 'timed_region_ms_at_least': MIN_MS, 'batch': B, 'dataset_complexes': len(pool),
           'steps_per_replay': S, 'capacities': {'cells': list(sb.cap_cells), 'complexes': B}, 'static_batch_mode': mode,
           'host_work_per_step': 'one hipGraph replay per %d steps (a StaticBatch of %d slots: the fill launches cut the tables, '
                                 'arrays and item tables of %d batches at once); the epoch\'s permutation is uploaded once per '
                                 'epoch, inside the timed region' % (S, S, S),
           'scope': 'device-side collate from the HBM-resident packed dataset + segment tables + item tables + the scope itself, '
                    'every step a batch of the shuffled epoch never seen before (PackedLoader, shuffle=True)'}
    all_fit = all(bool(sb.fits(epoch(e)).all()) for e in range(2 + EPOCHS))

    def leg(name, fn):
        try:
            out[name] = fn()
        except Exception as e:
            import traceback
            traceback.print_exc(file=sys.stderr)
            out[name] = {'failed': f'{type(e).__name__}: {e}'}
            torch.cuda.synchronize()

    def leg_propagate():
        g, keep = graph_of(prop_steps)
        # (an epoch that is not a multiple of S ends with a SHORTER captured sequence, as StaticForward / StaticTrainStep do)
        tail_n, g_tail = NB % S, None
        if tail_n:
            n_ = 1
            while n_ < tail_n:
                n_ <<= 1
            g_tail, keep_tail = graph_of(lambda: prop_steps(min(n_, S)))
            keep = (keep, keep_tail)

        def one(bs):
            sb.set_epoch(bs)
            for _ in range(len(bs) // S):
                g.replay()
            if len(bs) % S:
                (g_tail if (g_tail is not None and len(bs) % S == tail_n) else g).replay()
        cps, ms, n_ep = time_epochs(one)
        return {'cells_per_s': round(cps, 1), 'ms_per_step': round(ms, 5), 'epochs_timed': n_ep,
                'vs_fixed_batch_replay': round(cps / fixed_cells_per_s, 4) if fixed_cells_per_s else None}

    def leg_forward():
        from cwn_amd.static_graph import StaticForward
        sf = StaticForward(model, sb)
        with torch.no_grad():
            bs = epoch(1)
            sb.set_epoch(bs)
            outs = sf.replay()
            # the first batches of an epoch against the per-batch launches, bit for bit
            same = all(bool(torch.equal(outs[j][:len(bs[j])], model(packed.collate(bs[j])))) for j in range(min(S, 3)))
            # (an epoch that is not a multiple of S ends with a SHORTER replay -- StaticForward.slots_for -- instead of empty slots)
            def one(bs):
                sb.set_epoch(bs)
                k = 0
                while k < len(bs):
                    n = sf.slots_for(len(bs) - k)
                    sf.replay(n)
                    k += n
            cps, ms, n_ep = time_epochs(one)
        return {'cells_per_s': round(cps, 1), 'ms_per_step': round(ms, 5), 'epochs_timed': n_ep, 'bit_identical_to_per_batch_launches': same,
                'vs_fixed_batch_replay': round(fixed_forward_ms / ms, 4) if fixed_forward_ms else None}

    def leg_train():
        tmodel = copy.deepcopy(model).train()
        sb.set_epoch(epoch(0))
        ts = StaticTrainStep(tmodel, sb, task_type=task)
        ts.step()
        # (an epoch that is not a multiple of S ends with a SHORTER captured sequence -- StaticTrainStep.slots_for -- not with empty slots)
        cps, ms, n_ep = time_epochs(lambda bs: ts.run_epoch(bs, keep_losses=False))
        sb.set_epoch(epoch(1))
        finite = all(bool(torch.isfinite(l).item()) for l in ts.step())
        return {'cells_per_s': round(cps, 1), 'ms_per_step': round(ms, 5), 'epochs_timed': n_ep, 'loss_finite': finite,
                'vs_fixed_batch_replay': round(fixed_train_ms / ms, 4) if fixed_train_ms else None}

    def leg_fill():
        # what a never-seen batch costs BEFORE the scope runs: tables + capacity guard + collate (+ item tables, or -- mode 'csr' --
        # the CSR plans of every adjacency, which the fixed-batch forward / training legs find cached on their batch objects)
        g, keep = graph_of(lambda: sb.fill())
        sb.set_epoch(epoch(1))
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = max(4, 64 // S)
        for _ in range(n):
            sb.rewind(0)
            g.replay()
        torch.cuda.synchronize()
        return {'ms_per_step': round((time.perf_counter() - t0) / (n * S) * 1e3, 5),
                'launches': 'cwn_collate_tables + cwn_collate_guard + cwn_collate_slots + ' +
                            ('batched cwn_csr_build calls over all slots' if mode == 'csr' else 'cwn_layer_items_build_dev (forward [+ backward] tables)')}

    ROUTED = os.environ.get('CWN_BENCH_ROUTED') == '1' and mode == 'blocked'
    if ROUTED:
        # a dataset with complexes beyond one workgroup (molhiv's heavy tail): every epoch split between the blocked and the
        # csr-mode static batch (cwn_amd/static_graph.py: StaticRouter), one optimizer state
        from cwn_amd.static_graph import RoutedForward, RoutedTrainStep, StaticRouter

        def leg_forward_routed():
            router = StaticRouter(packed, B, slots=S)
            rf = RoutedForward(model, router)
            with torch.no_grad():
                a_, b_ = router.split(epoch(1))
                bs = epoch(1)
                outs = rf.run_epoch(bs)
                same = all(bool(torch.allclose(outs[k], model(packed.collate(bs[k])), rtol=0, atol=1e-5 * max(1.0, float(outs[k].abs().max()))))
                           for k in (list(a_[:2]) + list(b_[:2])))
                cps, ms, n_ep = time_epochs(rf.run_epoch)
            n_pool = int(sum(int((~rf.mask[np.asarray(ix)]).sum()) for ix in bs)) if rf.fbig is not None else 0
            return {'cells_per_s': round(cps, 1), 'ms_per_step': round(ms, 5), 'epochs_timed': n_ep, 'equal_to_per_batch_launches_1e-5': same,
                    'batches_without_a_complex_beyond_a_workgroup': len(a_), 'batches_with_one': len(b_),
                    # (round 6, RoutedForward(regroup=True): those batches keep the blocked path for the complexes that fit; the
                    #  others are pooled over the epoch into a csr-mode static batch of their own)
                    'regrouped': rf.fbig is not None, 'complexes_pooled_per_epoch': n_pool,
                    'vs_fixed_batch_replay': round(fixed_forward_ms / ms, 4) if fixed_forward_ms else None}

        def leg_train_routed():
            tmodel = copy.deepcopy(model).train()
            router = StaticRouter(packed, B, slots=S)
            rt = RoutedTrainStep(tmodel, router, task_type=task)
            cps, ms, n_ep = time_epochs(lambda bs: rt.run_epoch(bs, keep_losses=False))
            finite = all(bool(torch.isfinite(l).item()) for l in rt.run_epoch(epoch(1)) if l is not None)
            return {'cells_per_s': round(cps, 1), 'ms_per_step': round(ms, 5), 'epochs_timed': n_ep, 'loss_finite': finite,
                    'adam_steps': int(rt.opt.t), 'vs_fixed_batch_replay': round(fixed_train_ms / ms, 4) if fixed_train_ms else None}

        out['routed'] = True
        leg('forward', leg_forward_routed)
        leg('train', leg_train_routed)
        try:
            csr.check_errors(dev)
            out['device_error_word'] = 0
        except IndexError as e:
            out['device_error_word'] = str(e)
        out['every_batch_within_capacity'] = True       # (the router raises otherwise)
        return out
    LEGS = os.environ.get('CWN_BENCH_FRESH_LEGS', 'propagate,forward,train,fill').split(',')      # (profiling: one leg alone)
    for name_, fn_ in (('propagate', leg_propagate), ('forward', leg_forward), ('train', leg_train), ('fill', leg_fill)):
        if name_ in LEGS:
            leg(name_, fn_)
    try:
        f_ms = out['fill']['ms_per_step']
        for k, fixed in (('forward', fixed_forward_ms), ('train', fixed_train_ms)):
            if fixed and 'ms_per_step' in out.get(k, {}):
                out[k]['vs_fixed_batch_replay_plus_fill'] = round((fixed + f_ms) / out[k]['ms_per_step'], 4)
    except (KeyError, TypeError):
        pass
    all_fit = all_fit and all(bool(sb.fits(epoch(e)).all()) for e in range(2 + epochs_used[0]))      # (now incl. the backward table)
    out['every_batch_within_capacity'] = all_fit
    try:
        csr.check_errors(dev)
        out['device_error_word'] = 0
    except IndexError as e:
        out['device_error_word'] = str(e)
    return out
