#!/usr/bin/env python
"""bench.py -- cells/sec of the cellular message-passing hot path on N x MI355X.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): ZINC-like ring-lifted batch (max_ring 6), 128 complexes per
GPU, 4-layer EmbedSparseCIN (hidden 128, coboundary messages, edge embedding, BatchNorm;
exp/scripts/cwn-zinc.sh:14-30), synthetic inputs and random-init weights.

One STEP = one pass of the hot path over one batch with the batch already resident in HBM: for each
    of the 4 layers everything CochainMessagePassing.propagate does for the 3 cochain dimensions
    (12 propagate calls in the reference) as ONE complex-blocked launch (cwn_layer_fused_f32) straight
    from the int64 COO entries as delivered: the message function (coboundary Linear + ReLU), the
    gathers incl. the up_attr gather of data/complex.py:579-580, the scatter-adds, the zero fills,
    plus the GIN self terms.  The first layer's launch sorts the entries of each item into a per-item
    CSR and stores it; the other layers load it.  (Workloads the blocked launch does not serve --
    REDDIT hubs -- run a per-batch cwn_csr_build + grouped GEMM + cwn_aggregate_f32 instead.)
Unit: cell-updates/s = sum_d N_d x layers x steps / wall time, summed over ranks
(weak scaling: every rank owns its own batches, no data-path collective).
The JSON line also carries `roofline` (dominant kernel: layer_kernel<F, load>, csrc/cwn_layer.hip, at the configurations the
complex-blocked launch serves; aggregate_kernel on the CSR path), `cpu_baseline` (oracle timed on the host cores, rank 0,
N=1 only) and `secondary` (full model forward, training step, the same three scopes on batches never seen before, other
workloads).

Environment (exploration and the sub-runs of `secondary.workloads`; the driver sets none of them): CWN_BENCH_SKIP=leg,leg,...
(eager, concurrent, collate, train, fresh, workloads, roofline, full); CWN_BENCH_ATOMS=lo,hi | zinc (molecule sizes);
CWN_BENCH_MODEL=cinpp (EmbedCINpp on the ZINC workload); CWN_BENCH_MOLHIV_TAIL=p (share of 120 - 220-atom molecules) with
CWN_BENCH_ROUTED=1 (the never-seen-batch legs through StaticRouter); CWN_BENCH_DROPOUT=p (molhiv model; default 0.5 = the
reference's script); CWN_BENCH_FRESH_BATCHES / _SLOTS / _EPOCHS; CWN_BENCH_FORCE_DP=1 (world-1 RCCL group, the data-parallel
form of the training step forced) with CWN_BENCH_TRAIN_GRAPH=1; CWN_BENCH_SHARE_GPU=1 (N ranks over gloo on one GPU: control
flow only); CWN_BENCH_TRACE, CWN_BENCH_OVERLAP, CWN_BENCH_MIN_REGION_S, CWN_BENCH_DP_DEADLINE_S.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md); 6290 measured-achievable


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=200)
    p.add_argument('--warmup', type=int, default=20)
    p.add_argument('--workload', choices=['zinc', 'molhiv', 'reddit'], default='zinc',
                   help='zinc = BASELINE configs[1] (the headline); molhiv = configs[2]; reddit = configs[4]')
    p.add_argument('--batch', type=int, default=None, help='complexes per GPU per step (default: 128 / 512 / 32)')
    p.add_argument('--hidden', type=int, default=None)
    p.add_argument('--layers', type=int, default=None)
    p.add_argument('--num-batches', type=int, default=4, help='distinct synthetic batches cycled')
    p.add_argument('--cpu-seconds', type=float, default=12.0, help='budget of the CPU baseline leg')
    p.add_argument('--no-cpu', action='store_true')
    p.add_argument('--no-graph', action='store_true', help='eager launches instead of hipGraph replay')
    p.add_argument('--kernel-reps', type=int, default=100)
    p.add_argument('--only-primary', action='store_true', help='skip secondary / roofline / cpu legs (profiling)')
    p.add_argument('--brief', action='store_true', help='primary + roofline only (what the default run asks of the other workloads)')
    return p.parse_args()


def layer_algorithmic_bytes(stats, F, coboundary=True, out_streams=2):
    """SURVEY.md §8d, per conv layer (all three propagate calls): int64 indices as delivered,
    gather-counted fp32 rows, `out_streams` output streams per dimension (2 for SparseCIN; 3 for CIN++: up, down, boundaries)."""
    total = 0
    for d in range(3):
        e_up, b, n = stats[f'E_up{d}'], stats[f'B{d}'], stats[f'N{d}']
        total += e_up * (16 + 4 * F) + b * (16 + 4 * F) + 4 * F * n * out_streams
        if coboundary:
            total += e_up * (8 + 4 * F)
    return total


# see cwn_amd/train.py: captures must not be invalidated by the RCCL watchdog thread (N > 1)
CAPTURE_MODE = 'thread_local'


L2_TO_CU_BYTES_PER_US = 77.5e3  # measured, per CU, independent of how many CUs stream: (2304 - 576) KB in 31.7 - 9.4 us (profiles/r3_l2_stream.txt)
MFMA_F32_PEAK_TF = 157.3       # MI355X_MICROARCH.md: fp32-input MFMA, dense
MFMA_BF16_PEAK_TF = 2500.0     # MI355X_MICROARCH.md: bf16 MFMA, dense (never the 2:1-sparsity figure)


def load_profile_json(stem):
    """profiles/r<N>_<stem>.json of the latest round that has one (PMC traffic of the kernels, tools/collect_traffic.py)."""
    for r in (6, 5, 4, 3, 2):
        path = os.path.join(ROOT, 'profiles', f'r{r}_{stem}.json')
        if os.path.exists(path):
            with open(path) as fh:
                return json.load(fh)
    raise OSError(stem)


def gemm_roofline(flops, us, split, io_bytes, narrow, traffic):
    """Roofline entry of the grouped message GEMM.  The exact kernel is priced against the fp32-MFMA
    peak.  The split kernel (cwn_gemm_split.hip) issues SIX bf16 MFMAs per fp32-accurate product
    term, so its matrix-pipe ceiling is bf16 peak / 6 = 417 TFLOP/s of fp32-equivalent work; with
    N = K = 128 that floor (79 ps per row) lies BELOW the HBM floor of reading X and writing Y once
    (1 KiB per row: 128 ps), so the kernel is priced against HBM and the matrix-pipe figures are
    given beside it."""
    tf = flops / (us * 1e-6) / 1e12 if us > 0 else 0.0
    if not split:
        return {'bound': 'mfma', 'kernel': f'gemm_kernel ({"64x64" if narrow else "32x128"} tiles; grouped fp32-MFMA '
                                           'GEMM: coboundary-message products Y1, Y2)',
                'achieved': round(tf, 2), 'peak': MFMA_F32_PEAK_TF, 'unit': 'TFLOP/s',
                'frac': round(tf / MFMA_F32_PEAK_TF, 4), 'traffic': traffic, 'algorithmic_flops_per_launch': int(flops)}
    gbs = io_bytes / (us * 1e-6) / 1e9 if us > 0 else 0.0
    eq_peak = MFMA_BF16_PEAK_TF / 6.0
    return {'bound': 'hbm', 'kernel': 'gemm_split_kernel (64x128 tiles; grouped GEMM on the bf16 matrix pipe, exact '
                                      '3-way operand split, fp32 accuracy: coboundary-message products Y1, Y2)',
            'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(gbs / HBM_PEAK_GBS, 4),
            'traffic': traffic, 'algorithmic_bytes_per_launch': int(io_bytes),
            'algorithmic_flops_per_launch': int(flops), 'fp32_equivalent_tflops': round(tf, 2),
            'matrix_pipe_ceiling_tflops': round(eq_peak, 1), 'frac_of_matrix_pipe_ceiling': round(tf / eq_peak, 4),
            'frac_of_fp32_mfma_peak_157': round(tf / MFMA_F32_PEAK_TF, 4)}


def gen_complexes(spec):
    """One synthetic batch from a spec (kind, batch, seed, kwargs): what `gen(seed)` of main() returns."""
    kind, batch, seed, kw = spec
    from cwn_amd import synthetic as S
    if kind == 'zinc':
        return S.zinc_like_complexes(batch, seed, 6, **kw)
    if kind == 'molhiv':
        return S.molhiv_like_complexes(batch, seed, 6, **kw)
    return S.reddit_like_complexes(batch, seed)


LINE_BUDGET = 6144          # bytes: the driver keeps a bounded tail of stdout and parses ONE line (round 5's 22 KB line was lost)


def _pick(d, keys):
    return None if not isinstance(d, dict) else {k: d[k] for k in keys if d.get(k) is not None}


def _short(s, n):
    return s if not isinstance(s, str) or len(s) <= n else s[: n - 1] + '~'


def _rocprof_avg_us(full):
    """Average duration of the roofline kernel in the committed rocprofv3 --kernel-trace --stats summary of the same
    command (profiles/r*_kernel_avg.json, written by tools/collect_kernel_avg.py from the .md summaries); None when no
    summary covers this workload."""
    try:
        cfg = full.get('config') or {}
        tj = load_profile_json('kernel_avg')
        return (tj.get('entries') or {}).get(cfg.get('key'), {}).get('avg_us')
    except (OSError, ValueError, KeyError, AttributeError):
        return None


def slim_line(full, detail_path=None):
    """The ONE stdout line: the contract's keys + `roofline`, `cpu_baseline`, scalars of the secondary legs.  Everything
    else lives in `detail_path` (bench_detail.json) and on stderr.  Always shorter than LINE_BUDGET bytes: the optional
    blocks are dropped one by one (least important first) should a future leg outgrow it."""
    rf = full.get('roofline') or None
    roofline = None
    if rf:
        roofline = _pick(rf, ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'avg_launch_us',
                              'algorithmic_bytes_per_launch', 'compulsory_bytes_per_launch', 'frac_vs_pmc_traffic',
                              'launches_per_step', 'share_of_step', 'residency'))
        roofline['traffic'] = rf.get('traffic')           # (null = no PMC pass covers this configuration: stated, not dropped)
        roofline['kernel'] = _short(rf.get('kernel'), 96)
        roofline['avg_launch_us_source'] = 'hip events (graph replay of back-to-back launches, this run)'
        rp = _rocprof_avg_us(full)
        if rp and rf.get('algorithmic_bytes_per_launch'):
            roofline['avg_launch_us_rocprof'] = rp
            roofline['frac_rocprof'] = round(rf['algorithmic_bytes_per_launch'] / (rp * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
        roofline['traffic_source'] = 'profiles/*_traffic.json (committed PMC pass, same kernel + batch)'
    cb = full.get('cpu_baseline') or None
    cpu = None
    if cb:
        cpu = _pick(cb, ('value', 'unit', 'cores', 'kind', 'cpu_model', 'gpu_over_cpu', 'full_forward_cells_per_s',
                         'train_fwd_bwd_cells_per_s'))
        cpu['sample'] = _short(cb.get('sample'), 200)
    cfg = _pick(full.get('config'), ('workload', 'batch_per_gpu', 'hidden', 'layers', 'cells_per_batch', 'N', 'E_up', 'B',
                                     'launch', 'layer_kernel', 'parallelism'))
    if cfg:
        cfg['workload'] = _short(cfg.get('workload'), 200)
        cfg['layer_kernel'] = _short(cfg.get('layer_kernel'), 80)
    sec = full.get('secondary') or {}
    fb = sec.get('fresh_batches') or {}
    secondary = {
        'full_forward_ms': sec.get('full_forward_ms'),
        'full_forward_cells_per_s': sec.get('full_forward_cells_per_s'),
        'forward_breakdown_us': sec.get('forward_breakdown'),
        'train_step_ms': (sec.get('train_step') or {}).get('ms_per_step'),
        'train_step_cells_per_s': (sec.get('train_step') or {}).get('cells_per_s'),
        'eager_ms_per_step': (sec.get('eager_launches') or {}).get('ms_per_step'),
        'eager_full_forward_ms': (sec.get('eager_launches') or {}).get('full_forward_ms'),
        'concurrent_streams_cells_per_s': (sec.get('concurrent_streams') or {}).get('cells_per_s'),
        'collate_device_ms': (sec.get('collate') or {}).get('device_ms_per_batch'),
        'collate_cpu_oracle_ms': (sec.get('collate') or {}).get('cpu_oracle_ms_per_batch'),
        'fresh_batches': {k: _pick(fb.get(k), ('cells_per_s', 'ms_per_step', 'vs_fixed_batch_replay'))
                          for k in ('propagate', 'forward', 'train') if isinstance(fb.get(k), dict)} or None,
    }
    wls = {}
    for name, w in (sec.get('workloads') or {}).items():
        if not isinstance(w, dict):
            continue
        if 'failed' in w:
            wls[name] = {'failed': _short(w['failed'], 80)}
            continue
        e = {'value': w.get('value'), 'ms_per_step': w.get('ms_per_step'),
             'frac': (w.get('roofline') or {}).get('frac'), 'frac_step': (w.get('roofline_step') or {}).get('frac'),
             'forward_ms': w.get('full_forward_ms'), 'train_ms': w.get('train_step_ms')}
        f2 = w.get('fresh_batches') or {}
        for k in ('propagate', 'forward', 'train'):
            v = (f2.get(k) or {}).get('vs_fixed_batch_replay') if isinstance(f2.get(k), dict) else None
            if v is not None:
                e['fresh_' + k] = v
        wls[name] = {k: v for k, v in e.items() if v is not None}
    secondary['workloads'] = wls or None
    secondary = {k: v for k, v in secondary.items() if v is not None}
    out = {k: full.get(k) for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
                                    'scaling', 'vs_baseline', 'dtype', 'data')}
    out['config'] = cfg
    out['roofline'] = roofline
    out['roofline_step'] = _pick(full.get('roofline_step'), ('bound', 'achieved', 'peak', 'unit', 'frac', 'algorithmic_bytes_per_step'))
    out['roofline_mlp'] = _pick(full.get('roofline_mlp'), ('bound', 'achieved', 'peak', 'unit', 'frac', 'avg_launch_us'))
    out['cpu_baseline'] = cpu
    mg = full.get('multi_gpu') or {}
    out['multi_gpu'] = _pick(mg, ('rccl_ranks', 'backend', 'train_ms_per_step', 'exposed_allreduce_ms_per_step', 'bucket_bytes'))
    out['timing'] = _pick(full.get('timing'), ('rounds', 'timed_steps', 'timed_region_ms', 'ms_per_step_single_K_step_region'))
    out['secondary'] = secondary
    out['leg_seconds'] = full.get('leg_seconds')
    out['detail'] = detail_path
    for drop in ('leg_seconds', 'timing', 'roofline_mlp', ('secondary', 'workloads'), ('secondary', 'fresh_batches'), 'secondary'):
        if len(json.dumps(out)) < LINE_BUDGET:
            break
        if isinstance(drop, tuple):
            (out.get(drop[0]) or {}).pop(drop[1], None)
        else:
            out.pop(drop, None)
    return out


def emit(full):
    """Detail to bench_detail.json (beside this file; CWN_BENCH_DETAIL overrides) and to stderr, the slim line to stdout."""
    path = os.environ.get('CWN_BENCH_DETAIL', os.path.join(ROOT, 'bench_detail.json'))
    try:
        with open(path, 'w') as fh:
            json.dump(full, fh, indent=1)
        shown = os.path.relpath(path, ROOT)
    except OSError as e:
        print(f'[bench] could not write {path}: {e}', file=sys.stderr)
        shown = None
    print('[bench detail] ' + json.dumps(full), file=sys.stderr, flush=True)
    print(json.dumps(slim_line(full, shown)), flush=True)


from bench_fresh import fresh_batches_leg      # (the never-seen-batch legs: bench_fresh.py)


def self_launch(n):
    """Re-exec this command line under torch.distributed.run with n ranks on this node (127.0.0.1 rendezvous on a free
    port); the children see WORLD_SIZE and run main() as ranks.  Returns the launcher's exit code."""
    import socket
    import subprocess
    if not os.environ.get('CWN_BENCH_SHARE_GPU') == '1' and torch.cuda.device_count() < n:
        raise SystemExit(f'bench.py: --gpus {n} but this node shows {torch.cuda.device_count()} GPU(s) '
                         '(CWN_BENCH_SHARE_GPU=1 runs the ranks on one GPU over gloo: control flow only)')
    s_ = socket.socket()
    s_.bind(('127.0.0.1', 0))
    port = s_.getsockname()[1]
    s_.close()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print('[bench] --gpus %d without WORLD_SIZE: %s' % (n, ' '.join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    LEGS = {}
    t_leg = [time.perf_counter()]

    def mark(name):
        """wall seconds since the previous mark, filed under `name` (the line's `leg_seconds`: where a default run goes)"""
        now = time.perf_counter()
        LEGS[name] = LEGS.get(name, 0.0) + now - t_leg[0]
        t_leg[0] = now
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` by itself: launch the N ranks the contract describes (one process per GPU over RCCL)
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} '
                         f'(or plain `python bench.py --gpus {args.gpus}`, which starts the ranks itself)')
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert torch.cuda.is_available(), 'bench.py needs a GPU (the product path has no CPU fallback)'
    # CWN_BENCH_SHARE_GPU=1 (testing the multi-rank control flow on a one-GPU box): ranks share
    # device 0 and talk over gloo.  Never set by the driver: one rank per GPU over RCCL otherwise.
    share = os.environ.get('CWN_BENCH_SHARE_GPU') == '1'
    if share:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        import torch.distributed as dist
        if share:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)
    elif os.environ.get('CWN_BENCH_FORCE_DP') == '1':
        # the data-parallel form on the ONE GPU of a test box: a process group of one rank over RCCL, the training leg's
        # backward cut into chunks with an all-reduce each (cwn_amd/dist.py: FORCE_DP) -- `multi_gpu.exposed_allreduce_ms_per_step`
        # is then a real RCCL number (the buffer reduced with itself), everything but the xGMI wire
        import socket
        import torch.distributed as dist
        s_ = socket.socket()
        s_.bind(('127.0.0.1', 0))
        port_ = s_.getsockname()[1]
        s_.close()
        dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port_}', rank=0, world_size=1, device_id=dev)
        from cwn_amd import dist as _cd
        _cd.FORCE_DP = True
    else:
        dist = None

    # what a first multi-GPU run needs to read off the line: how many ranks talk over what, which device each one holds
    if dist is not None:
        world = dist.get_world_size()       # what the line reports is what the group holds
    MULTI = {'rccl_ranks': world, 'backend': None if dist is None else dist.get_backend(),
             'forced_data_parallel_form_on_one_rank': bool(world == 1 and dist is not None),
             'transport': None if dist is None else ('gloo over one shared GPU (CWN_BENCH_SHARE_GPU: control-flow test)' if share
                                                     else 'RCCL (torch.distributed "nccl") over xGMI, one process per GPU')}
    try:
        mine = {'rank': rank, 'local_rank': local_rank, 'device': torch.cuda.current_device(),
                'name': torch.cuda.get_device_name(dev), 'pid': os.getpid(),
                'hsa_ipc_legacy': os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}
        if dist is not None:
            allr = [None] * world
            dist.all_gather_object(allr, mine)
            MULTI['ranks'] = allr
        else:
            MULTI['ranks'] = [mine]
    except Exception as e:
        MULTI['ranks_error'] = f'{type(e).__name__}: {e}'

    from cwn_amd import _ffi, csr, ops
    from cwn_amd import layers as layers_mod
    from cwn_amd.complex import ComplexBatch
    from cwn_amd.models import EmbedCINpp, EmbedSparseCIN, OGBEmbedSparseCIN, SparseCIN
    CINPP = os.environ.get('CWN_BENCH_MODEL') == 'cinpp' and args.workload == 'zinc'
    from cwn_amd.synthetic import (batch_stats, zinc_like_complexes, molhiv_like_complexes,
                                   reddit_like_complexes)
    _ffi.lib()

    # workload = (model, generator) of one BASELINE config
    WL = args.workload
    defaults = {'zinc': (128, 128, 4), 'molhiv': (512, 64, 2), 'reddit': (32, 64, 4)}[WL]
    args.batch = args.batch or defaults[0]
    H, L = args.hidden or defaults[1], args.layers or defaults[2]
    args.hidden, args.layers = H, L
    torch.manual_seed(0)
    if WL == 'zinc':      # exp/scripts/cwn-zinc.sh:14-30
        # CWN_BENCH_MODEL=cinpp: the same stack over CINppConv layers (EmbedCINpp, mp/molec_models.py:167-199: three streams
        # per dimension, streaming path) -- the `zinc_cinpp` entry of secondary.workloads
        Model_ = EmbedCINpp if CINPP else EmbedSparseCIN
        model = Model_(28, 4, 1, L, H, dropout_rate=0.0, max_dim=2, jump_mode=None,
                       nonlinearity='relu', readout='sum', train_eps=False,
                       final_hidden_multiplier=2, final_readout='sum', init_reduce='sum',
                       embed_edge=True, use_coboundaries=True, graph_norm='bn')
        # CWN_BENCH_ATOMS=lo,hi (exploration, never the headline): molecule sizes other than the generator's 18 - 30 atoms,
        # e.g. 9,38 for the spread of the real ZINC subset (mixed launches / big items, docs/history/DESIGN_rounds1-5.md 4.0b-c)
        # ... or 'zinc': the size statistics of the real ZINC-12k subset (9 - 37 atoms, mean 23.2: cwn_amd/synthetic.py)
        if os.environ.get('CWN_BENCH_ATOMS') == 'zinc':
            GEN_KIND, GEN_KW = 'zinc', dict(size_dist='zinc')
        else:
            atoms = tuple(int(v) for v in os.environ.get('CWN_BENCH_ATOMS', '18,30').split(','))
            GEN_KIND, GEN_KW = 'zinc', dict(n_lo=atoms[0], n_hi=atoms[1])
        coboundary = True
    elif WL == 'molhiv':  # exp/scripts/cwn-molhiv.sh:9-32, batch per BASELINE.json
        # (--drop_rate 0.5 --indrop_rate 0.0 --drop_position lin2: after every conv layer and before lin2, in TRAINING mode --
        #  the training legs below; the eval-mode scopes are what they were)
        model = OGBEmbedSparseCIN(1, L, H, dropout_rate=float(os.environ.get('CWN_BENCH_DROPOUT', '0.5')), indropout_rate=0.0, max_dim=2,
                                  readout='mean', final_readout='sum', apply_dropout_before='lin2',
                                  init_reduce='sum', embed_edge=True, use_coboundaries=True, graph_norm='bn')
        # CWN_BENCH_MOLHIV_TAIL (the `molhiv_real_tail` entry of secondary.workloads: 5e-4): the dataset's molecules of 120 - 220 atoms
        MTAIL = float(os.environ.get('CWN_BENCH_MOLHIV_TAIL', '0'))
        GEN_KIND, GEN_KW = 'molhiv', dict(tail=MTAIL)
        coboundary = True
    else:                 # exp/scripts/mpsn-redditb.sh:6-28
        model = SparseCIN(1, 2, L, H, dropout_rate=0.0, max_dim=2, jump_mode='cat', readout='sum',
                          use_coboundaries=False, graph_norm='id')
        with torch.no_grad():
            for p_ in model.parameters():
                p_.mul_(0.3)      # no norm layer and degrees up to 300: keep activations finite
        GEN_KIND, GEN_KW = 'reddit', {}
        coboundary = False
    def gen(seed):
        return gen_complexes(gen.spec(seed))
    gen.spec = lambda seed: (GEN_KIND, args.batch, seed, GEN_KW)
    model = model.to(dev).eval()
    # the criterion of the training legs: exp/scripts/cwn-zinc.sh --task_type regression (L1), cwn-molhiv.sh bin_classification
    # (BCE with logits), mpsn-redditb.sh classification (CrossEntropyLoss, exp/train_utils.py:21-22)
    TASK = {'zinc': 'regression', 'molhiv': 'bin_classification', 'reddit': 'classification'}[WL]
    DROP = float(getattr(model, 'dropout_rate', 0.0)) if getattr(model, 'conv_dropout', False) else 0.0

    # ---- synthetic batches, resident in HBM ---------------------------------------------------
    fixed = [gen(1000 * rank + i) for i in range(args.num_batches)]
    cpu_batches = [ComplexBatch.from_complex_list(cs, max_dim=2) for cs in fixed]
    stats = [batch_stats(b) for b in cpu_batches]
    types = [tuple(None if b.cochains[d].x is None else b.cochains[d].x.clone() for d in range(3))
             for b in cpu_batches]
    batches = [ComplexBatch.from_complex_list(cs, max_dim=2).to(dev) for cs in fixed]   # .to() is in place
    del fixed
    types_dev = [tuple(None if t is None else t.to(dev) for t in ts) for ts in types]

    def reset_inputs(bi):
        """model(b) overwrites the container's features; put the raw input features back."""
        b = batches[bi]
        for d in range(3):
            b.cochains[d]._x = types_dev[bi][d]
        return b

    # per-layer input features of every batch (one full forward each), so the timed region can run
    # the propagate scope of every layer on the features that layer really sees
    layer_inputs = []
    with torch.no_grad():
        for bi in range(len(batches)):
            b = reset_inputs(bi)
            if hasattr(model, 'init_conv'):
                params = b.get_all_cochain_params(max_dim=2, include_down_features=False)
                x0 = [x.contiguous() for x in model.init_conv(*params)]
            else:
                x0 = [b.cochains[d].x.contiguous() for d in range(3)]
            _, res = model(reset_inputs(bi), include_partial=True)
            layer_inputs.append([x0] + [[res[f'layer{l - 1}_{d}'].contiguous() for d in range(3)]
                                        for l in range(1, L)])
    torch.cuda.synchronize()

    # CWN_BENCH_OVERLAP=1: the plan build on a side stream underneath the layer-0 message GEMMs, which
    # do not need it (csr.build_many(overlap=True)).  Off by default: inside a hipGraph the fork/join
    # costs more than the 10 us it hides (measured: ZINC-128 80.8 -> 89.6 us/step, MOLHIV 83.9 -> 86.5,
    # REDDIT-like and batch 8192 unchanged); it pays in eager mode only.
    OVERLAP_PLAN_BUILD = os.environ.get('CWN_BENCH_OVERLAP', '0') == '1'

    # Does this configuration run the complex-blocked layer kernel (csrc/cwn_layer.hip: one launch per
    # layer straight from the int64 COO indices, no CSR plan)?  Probed once, outside the timed region.
    with torch.no_grad():
        b0 = batches[0]
        b0.set_xs(layer_inputs[0][0])
        probe_plans, _ = model.convs[0].propagate_all(*b0.get_all_cochain_params(max_dim=2, include_down_features=False))
        BLOCKED = probe_plans[0] == 'blocked'
        FORM = None
        if BLOCKED:
            t_ = model.convs[0]._blocked_args(b0.get_all_cochain_params(max_dim=2, include_down_features=False), 0)[2]
            FORM = {'variant': t_.variant, 'items_per_launch': t_.n_items, 'big_items': int(getattr(t_, 'n_big', 0)),
                    'form': {0: '16 waves, one workgroup per CU', 1: '8 waves <= 128 VGPRs <= 80 KiB LDS, two workgroups per CU',
                             'mixed': 'two launches: two-per-CU form for the complexes that fit it + 16-wave form for the rest'}[t_.variant]}
        if rank == 0:
            print(f'[bench] complex-blocked layer kernel: {BLOCKED}'
                  + ('' if BLOCKED else f' ({model.convs[0].blocked_reason})'), file=sys.stderr)
        csr._cache.clear()
    torch.cuda.synchronize()

    def propagate_scope(bi):
        """One step: (CSR path only: fresh plans for the batch, then) the propagate scope of every layer."""
        b, feats = batches[bi], layer_inputs[bi]
        if BLOCKED:
            b.block_plan().forget_csr()      # a step starts from the int64 COO entries, like a new batch
        else:
            csr._cache.clear()
            b.prepare(max_dim=2, overlap=OVERLAP_PLAN_BUILD)
        outs = None
        for l, conv in enumerate(model.convs):
            b.set_xs(feats[l])
            params = b.get_all_cochain_params(max_dim=2, include_down_features=False)
            _, outs = conv.propagate_all(*params)
        return outs

    def full_forward(bi):
        return model(reset_inputs(bi))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps, warmup, use_graph, min_seconds=0.0):
        """W warm-up steps, then timed regions of exactly K steps (see `region`); times are the max over ranks."""
        nb = len(batches)
        graphs = None
        with torch.no_grad():
            if use_graph:
                # one hipGraph per distinct batch: the step's launches are captured from the very
                # same C-ABI calls (stream capture), replay removes the Python/launch overhead
                graphs = []
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    for bi in range(nb):
                        fn(bi)
                torch.cuda.current_stream().wait_stream(s)
                torch.cuda.synchronize()
                for bi in range(nb):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
                        keep = fn(bi)
                    graphs.append((g, keep))
                # a graph launch costs ~9 us of host/queue time whatever it holds, so the steady-state
                # loop replays ONE graph that holds a step of every distinct batch (nb steps)
                g_all = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_all, capture_error_mode=CAPTURE_MODE):
                    keep_all = [fn(bi) for bi in range(nb)]
                # ... and, when the timed region is long enough, one that holds several such rounds (the gap
                # between two graph launches is ~8 us of GPU time: 5 % of a 4-step graph of the blocked path)
                # (small batches only: a captured step keeps its output tensors alive)
                # one round fewer than the timed region holds: a SMALL graph goes first (a graph's packets are
                # written before its first kernel starts: ~0.5 us a node), the large ones are enqueued behind it
                rounds = min(steps - nb, 32) // nb if max(st_['cells'] for st_ in stats) <= 50_000 else 0
                g_big = None
                if rounds >= 2:
                    g_big = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g_big, capture_error_mode=CAPTURE_MODE):
                        keep_big = [fn(bi) for _ in range(rounds) for bi in range(nb)]

            def run_steps(n_steps, start=0):
                """exactly n_steps steps, batches cycled"""
                if not use_graph:
                    for i in range(n_steps):
                        fn((start + i) % nb)
                    return
                i = 0
                while i < n_steps and (start + i) % nb != 0:       # align to batch 0
                    graphs[(start + i) % nb][0].replay()
                    i += 1
                if g_big is not None and n_steps - i >= (rounds + 1) * nb:
                    g_all.replay()                                 # short enqueue: the GPU starts early
                    i += nb
                while g_big is not None and n_steps - i >= rounds * nb:
                    g_big.replay()
                    i += rounds * nb
                while n_steps - i >= nb:
                    g_all.replay()
                    i += nb
                while i < n_steps:
                    graphs[(start + i) % nb][0].replay()
                    i += 1

            def region(n_rounds):
                """n_rounds x exactly `steps` steps between barrier + synchronize on both sides: wall seconds, and
                the duration of every round from HIP events recorded behind it on the launch stream"""
                evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_rounds + 1)]
                barrier()
                t0 = time.perf_counter()
                evs[0].record()
                for r in range(n_rounds):
                    run_steps(steps, start=warmup)
                    evs[r + 1].record()
                # poll for the end of the stream before the (blocking) synchronize of barrier(): a blocked host
                # thread is woken tens of microseconds after the GPU is done -- 5 % of a 20-step timed region
                while not evs[-1].query():
                    pass
                barrier()
                wall = time.perf_counter() - t0
                return wall, [evs[r].elapsed_time(evs[r + 1]) for r in range(n_rounds)]

            def max_over_ranks(v):
                if dist is None:
                    return v
                t = torch.tensor([v], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                return float(t.item())

            run_steps(warmup)
            # the contract's literal region: exactly K steps ...
            k_wall, _ = region(1)
            k_wall = max_over_ranks(k_wall)
            rounds_ = 1
            if min_seconds > 0:
                # ... and, because K steps of this path are under a millisecond (one graph launch + one host wake-up
                # are ~8 % of it: VERDICT r2 weak #10), the same K steps R times back to back in ONE region of at least
                # `min_seconds`; R is agreed across the ranks (it follows from the slowest rank's K-step time)
                rounds_ = int(min(4096, max(1, -(-min_seconds // max(k_wall, 1e-6)))))
            wall, round_ms = region(rounds_)
            wall = max_over_ranks(wall)
        return {'dt': wall, 'rounds': rounds_, 'round_ms': round_ms, 'k_region_s': k_wall}

    mark('setup')
    use_graph = not args.no_graph
    MIN_REGION_S = float(os.environ.get('CWN_BENCH_MIN_REGION_S', '0.05'))
    try:
        tm = timed(propagate_scope, args.steps, args.warmup, use_graph, MIN_REGION_S)
    except Exception as e:   # capture restrictions differ between ROCm builds: say so, run eager
        if not use_graph:
            raise
        print(f'[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eager', file=sys.stderr)
        use_graph = False
        torch.cuda.synchronize()
        tm = timed(propagate_scope, args.steps, args.warmup, False, MIN_REGION_S)
    # the timed region is R rounds of exactly K steps; everything below is per K steps
    R = tm['rounds']
    dt = tm['dt'] / R
    round_ms = sorted(tm['round_ms'])
    timing = {'rounds': R, 'timed_steps': R * args.steps, 'timed_region_ms': round(tm['dt'] * 1e3, 3),
              'ms_per_step_round_median': round(round_ms[len(round_ms) // 2] / args.steps, 5),
              'ms_per_step_round_min': round(round_ms[0] / args.steps, 5),
              'ms_per_step_round_max': round(round_ms[-1] / args.steps, 5),
              'ms_per_step_single_K_step_region': round(tm['k_region_s'] / args.steps * 1e3, 5),
              'note': f'`value` and `ms_per_step` = wall clock over ONE region of {R} rounds x exactly {args.steps} steps '
                      '(barrier + synchronize on both sides, max over ranks); per-round figures from HIP events behind '
                      'every round on the launch stream; the single K-step region (the same bracket around K steps '
                      'only) carries one graph launch and one host wake-up per K steps'}

    cells_per_step_local = sum(stats[(args.warmup + i) % len(stats)]['cells'] for i in range(args.steps)) * L / args.steps
    cells_total = torch.tensor([cells_per_step_local * args.steps], device=dev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(cells_total)
    value = float(cells_total.item()) / dt

    mark('primary')
    # secondary: the full model forward (embedding, 4 conv layers incl. MLPs/BN, readout, head)
    SKIP = set(filter(None, os.environ.get('CWN_BENCH_SKIP', '').split(',')))   # debugging: legs to skip
    if args.brief:
        args.no_cpu = True
        SKIP |= {'full', 'eager', 'concurrent', 'train', 'collate', 'workloads', 'fresh'}
    if args.only_primary:
        args.no_cpu = True
    try:
        if args.only_primary or 'full' in SKIP:
            raise KeyboardInterrupt
        dt_full = timed(full_forward, max(args.steps // 4, 10), max(args.warmup // 4, 3), use_graph)['dt']
    except KeyboardInterrupt:
        dt_full = float('nan')
    except Exception as e:
        print(f'[bench] full-forward graph capture failed ({type(e).__name__}); eager', file=sys.stderr)
        torch.cuda.synchronize()
        dt_full = timed(full_forward, max(args.steps // 4, 10), max(args.warmup // 4, 3), False)['dt']
    mark('full_forward')
    # secondary: the SAME propagate-scope step launched eagerly (Python + ctypes per launch, no graph):
    # what a caller pays today when every batch has new shapes and nothing can be replayed
    eager = None
    if use_graph and not args.only_primary and 'eager' not in SKIP:
        try:
            esteps = max(args.steps // 4, 10)
            te_ = timed(propagate_scope, esteps, 3, False, MIN_REGION_S)      # (regions of >= 50 ms, as the headline's)
            dte = te_['dt'] / te_['rounds']
            ecells = torch.tensor([sum(stats[(3 + i) % len(stats)]['cells'] for i in range(esteps)) * L],
                                  device=dev, dtype=torch.float64)
            if dist is not None:
                dist.all_reduce(ecells)
            from cwn_amd import _cext
            eager = {'cells_per_s': round(float(ecells.item()) / dte, 1), 'ms_per_step': round(dte / esteps * 1e3, 5),
                     'binding': _cext.active(),
                     'note': ('same step through the reference-shaped API (set_xs, get_all_cochain_params, conv): one '
                              'Python call per layer launch, the per-call part of a prepared launch in C++ '
                              '(cwn_amd/_cwn_torch_ext.so; ctypes when it is not built)' if BLOCKED else
                              'same step, one Python call per launch and a validated (synchronising) plan build')
                             + ': host-bound; the headline replays the step from a hipGraph'}
            # ... and the whole model eagerly, model(batch) (r4: 0.91 ms, 8 x the replay; prepared launches + one range check
            # per forward in round 5)
            fsteps = max(args.steps // 4, 10)
            tf_ = timed(full_forward, fsteps, 3, False, MIN_REGION_S)
            dtf = tf_['dt'] / tf_['rounds']
            eager['full_forward_ms'] = round(dtf / fsteps * 1e3, 5)
        except Exception as e:
            print(f'[bench] eager leg failed: {type(e).__name__}: {e}', file=sys.stderr)
    full_steps = max(args.steps // 4, 10)
    full_cells = torch.tensor([sum(stats[i % len(stats)]['cells'] for i in range(full_steps)) * L],
                              device=dev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(full_cells)

    mark('eager')
    # ---- rooflines: both kernels of a layer are measured live; the one with the larger share of the
    # step is `roofline` (dominant), the other `roofline_other` ---------------------------------------
    roofline = roofline_other = r_plan = roofline_mlp = None
    forward_breakdown = {}
    if rank == 0 and not args.only_primary and 'roofline' not in SKIP:

        def replay_us(fn, reps):
            """Average duration of back-to-back dependent launches replayed from a hipGraph, between
            two HIP events on the replay stream (the host -- Python/ctypes, ~4 us per call -- is out
            of the measurement; the ~1.5 us dependent-launch boundary is in)."""
            fn()
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fn()
            torch.cuda.current_stream().wait_stream(side)
            kg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(kg, capture_error_mode=CAPTURE_MODE):
                for _ in range(reps):
                    fn()
            kg.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                kg.replay()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / (5 * reps)

        b, feats = batches[0], layer_inputs[0]
        step_us = dt / args.steps * 1e6
        alg = layer_algorithmic_bytes(stats[0], H, coboundary=coboundary, out_streams=3 if CINPP else 2)
        if BLOCKED:
            # ONE kernel per layer: layer_kernel (csrc/cwn_layer.hip).  Launch 0 of a step reads and sorts
            # the COO entries and stores every item's CSR ("store"), launches 1.. load it back ("load").
            with torch.no_grad():
                b.set_xs(feats[1])
                params = b.get_all_cochain_params(max_dim=2, include_down_features=False)
                dims, plan_, table, _key = model.convs[1]._blocked_args(params, 0)
                ll_ = ops.LayerLaunch(dims, table)         # (one launch per item table; BIG records get their scratch)
                xs_ = [D_.x for D_ in dims]
                ll_.run(xs_, _ffi.LAYER_CSR_STORE)
                load_us = replay_us(lambda: ll_.run(xs_, _ffi.LAYER_CSR_LOAD), args.kernel_reps)
                store_us = replay_us(lambda: ll_.run(xs_, _ffi.LAYER_CSR_STORE), args.kernel_reps)
            s0_ = stats[0]
            gemm_rows = (s0_['N0'] + s0_['N1']) + (s0_['N1'] + s0_['N2'])      # Y1 | Y2 rows of both GEMM dimensions
            flops = 2.0 * gemm_rows * H * H
            # compulsory bytes: every feature row the launch needs once, the entries once, both outputs once
            compulsory = (4 * H * (gemm_rows + s0_['N0']) + 24 * (s0_['E_up0'] + s0_['E_up1'])
                          + 16 * (s0_['B1'] + s0_['B2']) + 2 * 4 * H * s0_['cells'])
            traffic = None
            try:
                tj = load_profile_json('traffic')
                ent = tj['entries'].get(str(args.batch) if WL == 'zinc' else f'{WL}:{args.batch}')
                if (WL != 'zinc' or tj.get('hidden') == H) and ent and ent.get('kernel', '').startswith(f'layer_kernel<{H}'):
                    traffic = ent['traffic_bytes']
            except (OSError, ValueError, KeyError):
                traffic = None
            gbs = alg / (load_us * 1e-6) / 1e9
            tf = flops / (load_us * 1e-6) / 1e12
            layer_us = ((L - 1) * load_us + store_us) / L
            roofline = {
                'bound': 'hbm',
                'kernel': f'layer_kernel<{H}, load> (complex-blocked ' + ('CIN++' if CINPP else 'SparseCIN') + ' propagate step: message GEMMs on the '
                          'bf16 matrix pipe into LDS, per-complex CSR, both reductions + self terms out of LDS)',
                'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(gbs / HBM_PEAK_GBS, 4),
                'traffic': traffic,
                'residency': 'L2/MALL' if args.batch * H <= 512 * 128 else 'HBM',
                'traffic_source': 'profiles/*_traffic.json (committed PMC pass of the same kernel and batch size; not re-measured in this run)',
                'frac_vs_pmc_traffic': None if traffic is None else round(traffic / (load_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                'algorithmic_bytes_per_launch': alg, 'compulsory_bytes_per_launch': int(compulsory),
                'avg_launch_us': round(load_us, 3), 'avg_launch_us_store_variant': round(store_us, 3),
                'launches_per_step': L, 'share_of_step': round(L * layer_us / step_us, 3),
                'frac_of_measured_achievable_6290': round(gbs / 6290.0, 4),
                'note': 'algorithmic bytes = SURVEY.md 8(d), gather-counted (a row is counted once per entry that '
                        'reads it); the kernel reads each row once per workgroup, so its real traffic (`traffic`, PMC) '
                        'is the compulsory figure plus the packed weights; avg over back-to-back dependent launches '
                        'replayed from a hipGraph between two HIP events.  `achieved` is a BYTE RATE quoted against the HBM '
                        'peak, not DRAM traffic: the cycled batches (~60 MB) and the weights live in L2 / MALL between '
                        'launches -- `traffic` (PMC, when the committed pass covers this batch size) is what the memory '
                        'side moved'}
            # the dense half of the layer: update_up_nn / update_boundaries_nn / combine_nn of all dimensions in one
            # launch (csrc/cwn_mlp.hip) -- the north star's MFMA target, priced against the matrix pipe
            try:
                with torch.no_grad():
                    outs_ = ops.layer_fused(dims, table, _ffi.LAYER_CSR_LOAD)
                    plans_ = ['blocked'] * 3
                    conv1 = model.convs[1]
                    if conv1._dense_eval(plans_, outs_, 0) is not None and layers_mod.FUSED_UPDATE_MLP:
                        mlp_us = replay_us(lambda: conv1._dense_eval(plans_, outs_, 0), args.kernel_reps)
                        mflops = 2.0 * s0_['cells'] * (9 if CINPP else 6) * H * H     # five Linear layers per cell, the combine 2H wide (CIN++: seven, 3H)
                        mtf = mflops / (mlp_us * 1e-6) / 1e12
                        roofline_mlp = {
                            'bound': 'mfma', 'kernel': (f'update_mlp3_kernel<{H}> (update_up_nn, update_down_nn, update_boundaries_nn, 3H-wide combine_nn' if CINPP else
                                                        f'update_mlp_kernel<{H}> (update_up_nn, update_boundaries_nn, combine_nn') + ' of all '
                                                       'dimensions in one launch; exact 3-way bf16 split, six MFMAs per product term)',
                            'achieved': round(mtf, 2), 'peak': round(MFMA_BF16_PEAK_TF / 6.0, 1), 'unit': 'TFLOP/s',
                            'frac': round(mtf / (MFMA_BF16_PEAK_TF / 6.0), 4), 'traffic': None,
                            'frac_of_fp32_mfma_peak_157': round(mtf / MFMA_F32_PEAK_TF, 4),
                            'algorithmic_flops_per_launch': int(mflops), 'avg_launch_us': round(mlp_us, 3),
                            'weight_stream_bytes_per_launch': int(-(-s0_['N0'] // (4096 // H)) + -(-s0_['N1'] // (4096 // H)) + -(-s0_['N2'] // (4096 // H))) * (9 if CINPP else 6) * H * H * 6,
                            'note': 'fp32-equivalent FLOPs (2 M N K per Linear) / launch time; every workgroup streams the six packed '
                                    'weights out of L2 (weight_stream_bytes_per_launch), which is what bounds it at this batch size'}
                        # what actually bounds it: a CU receives ~77.5 GB/s (~35 B per clock) from L2 whatever the other CUs do
                        # (profiles/r3_l2_stream.txt: 32 ... 256 workgroups streaming one 576-KB buffer, same or spread
                        # addresses, tools/proto/l2_stream.hip), and a workgroup needs its 6 weights + its two input tiles
                        wg_bytes = (9 if CINPP else 6) * H * H * 6 + (3 if CINPP else 2) * (4096 // H) * H * 4
                        floor_us = wg_bytes / L2_TO_CU_BYTES_PER_US
                        roofline_mlp['l2_to_cu_stream'] = {
                            'bytes_per_workgroup': wg_bytes, 'measured_cap_GB_per_s_per_CU': round(L2_TO_CU_BYTES_PER_US / 1e3, 1),
                            'floor_us': round(floor_us, 2), 'frac_of_floor': round(floor_us / mlp_us, 3),
                            'note': 'one workgroup per CU, one round: the launch cannot be faster than one workgroup\'s stream'}
            except Exception as e:
                print(f'[bench] update-mlp roofline failed: {type(e).__name__}: {e}', file=sys.stderr)
            # where a full forward goes, launch by launch (the same back-to-back replay measurement)
            try:
                with torch.no_grad():
                    bb = reset_inputs(0)
                    prm0 = bb.get_all_cochain_params(max_dim=2, include_down_features=False)
                    if hasattr(model, 'init_conv'):
                        front_us = replay_us(lambda: model.init_conv(*prm0), args.kernel_reps)
                        forward_breakdown['front_us'] = round(front_us, 3)
                    xs_ = [x.contiguous() for x in layer_inputs[0][L - 1]]
                    if model._head_fused(xs_, bb, False, {}) is not None:
                        forward_breakdown['head_us'] = round(replay_us(lambda: model._head_fused(xs_, bb, False, {}), args.kernel_reps), 3)
                    forward_breakdown['propagate_us_per_layer'] = round(load_us, 3)
                    if roofline_mlp is not None:
                        forward_breakdown['update_mlp_us_per_layer'] = roofline_mlp['avg_launch_us']
            except Exception as e:
                print(f'[bench] forward breakdown failed: {type(e).__name__}: {e}', file=sys.stderr)
            eq_peak = MFMA_BF16_PEAK_TF / 6.0
            roofline_other = {
                'bound': 'mfma', 'kernel': 'the same launch against the matrix pipe: six v_mfma_f32_16x16x32_bf16 per '
                                           'fp32-accurate product term (bf16 peak / 6)',
                'achieved': round(tf, 2), 'peak': round(eq_peak, 1), 'unit': 'TFLOP/s', 'frac': round(tf / eq_peak, 4),
                'traffic': None, 'algorithmic_flops_per_launch': int(flops),
                'frac_of_fp32_mfma_peak_157': round(tf / MFMA_F32_PEAK_TF, 4)}
        else:
            with torch.no_grad():
                csr._cache.clear()
                b.prepare(max_dim=2)
                b.set_xs(feats[1])
                params = b.get_all_cochain_params(max_dim=2, include_down_features=False)
                lv = model.convs[1].mp_levels
                gemms, owner = [], []
                for d in range(3):
                    sp = lv[d].gemm_specs(params[d])
                    gemms += sp
                    owner += [d] * len(sp)
                ys = ops.run_gemm(gemms, dev) if gemms else []
                streams = []
                for d in range(3):
                    mine = [y for y, o in zip(ys, owner) if o == d]
                    streams += lv[d].streams(params[d], mine or None)
                for st in streams:
                    st.validate()
                specs = []
                for st in streams:
                    s = ops.AggSpec(adj=st.adj, n_dst=st.n_dst, F=st.width, msg_op=st.msg_op,
                                    reduce=_ffi.REDUCE[st.reduce], self_x=st.self_x, eps=st.eps)
                    if st.adj is not None:
                        s.A, s.ia = st.A, st.adj.col
                        if st.msg_op != ops.MSG_A:
                            s.B, s.ib = st.B, st.adj.aux
                    specs.append(s)
                agg_us = replay_us(lambda: ops.run_aggregate(specs, dev), args.kernel_reps)
                gemm_us = replay_us(lambda: ops.run_gemm(gemms, dev), args.kernel_reps) if gemms else 0.0

                def rebuild_plans():
                    csr._cache.clear()
                    b.prepare(max_dim=2)
                plan_us = replay_us(rebuild_plans, max(args.kernel_reps // 4, 4))
                adjs = list(csr._cache.values())
            achieved = alg / (agg_us * 1e-6) / 1e9
            # HBM bytes per launch from the PMC passes committed under profiles/ (same kernel, same
            # workload shape); None when no pass exists for this configuration
            traffic = None
            try:
                tj = load_profile_json('traffic')
                ent = tj['entries'].get(str(args.batch) if WL == 'zinc' else f'{WL}:{args.batch}')
                if (WL != 'zinc' or tj.get('hidden') == H) and ent and ent.get('kernel', '').startswith('aggregate_kernel'):
                    traffic = ent['traffic_bytes']
            except (OSError, ValueError, KeyError):
                traffic = None
            r_agg = {'bound': 'hbm', 'kernel': 'aggregate_kernel<4> (fused gather-message-reduce, all dims of a layer)',
                     'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic,
                     'algorithmic_bytes_per_launch': alg, 'avg_launch_us': round(agg_us, 3),
                     'launches_per_step': L, 'share_of_step': round(L * agg_us / step_us, 3),
                     'frac_of_measured_achievable_6290': round(achieved / 6290.0, 4)}
            # K from the operands (W may be the whole [N, 2F] weight addressed through w_col0)
            # HBM bytes of the grouped GEMM from the same PMC passes (FETCH_SIZE doubled for 16-B/lane
            # streams as MI355X_MICROARCH.md prescribes, WRITE_SIZE as is), K <= 128 wide-tile kernel
            gemm_traffic = None
            try:
                raw = load_profile_json('pmc_fetch_write_raw').get(str(args.batch), {})
                if WL == 'zinc' and H == 128:
                    for kname, v in raw.items():
                        if kname.startswith('gemm_split_kernel') or (gemm_traffic is None and kname.startswith('gemm_kernel<true, false, 128, 4')):
                            gemm_traffic = int((2 * v['FETCH_SIZE_KB_avg'] + v['WRITE_SIZE_KB_avg']) * 1024)
            except (OSError, ValueError, KeyError):
                gemm_traffic = None
            flops = 2.0 * sum(g.X.size(0) * g.W.size(0) * (g.X.size(1) + (g.X2.size(1) if g.X2 is not None else 0))
                              for g in gemms)
            tf = flops / (gemm_us * 1e-6) / 1e12 if gemms else 0.0
            r_gemm = gemm_roofline(flops=flops, us=gemm_us, split=bool(gemms) and ops.gemm_uses_split(gemms, dev),
                                   io_bytes=sum(4 * (g.X.numel() + g.X.size(0) * g.W.size(0) + g.W.size(0) * g.X.size(1)
                                                     + (g.W.size(0) if g.bias is not None else 0)) for g in gemms),
                                   narrow=H <= 64, traffic=gemm_traffic)
            r_gemm.update({'avg_launch_us': round(gemm_us, 3), 'launches_per_step': L if gemms else 0,
                           'share_of_step': round(L * gemm_us / step_us, 3)})
            # plan build (cwn_csr_build): key + val (+ aux) int64 in, rowptr + col + perm (+ aux) int32 out
            plan_bytes = 0
            for ent in adjs:
                a = ent[2]          # csr._cache values are (version, weakref, Adjacency)
                has_aux = a.aux is not None
                plan_bytes += a.n_entries * (16 + 8 + (12 if has_aux else 0)) + 4 * (a.n_dst + 1)
            r_plan = {'bound': 'hbm', 'kernel': 'cwn_csr_build (destination-sorted int32 CSR of every adjacency of the batch, once per step)',
                      'achieved': round(plan_bytes / (plan_us * 1e-6) / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                      'frac': round(plan_bytes / (plan_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), 'traffic': None,
                      'algorithmic_bytes_per_launch': int(plan_bytes), 'avg_launch_us': round(plan_us, 3),
                      'launches_per_step': 1, 'share_of_step': round(plan_us / step_us, 3)}
            note = ('avg over back-to-back dependent launches replayed from a hipGraph between two HIP events '
                    '(includes the ~1.5 us inter-kernel boundary; rocprofv3 kernel-only averages are in profiles/, '
                    'where few-us kernels read ~1.5-3 us high); batch 128 is latency-bound and L2/MALL-resident')
            roofline, roofline_other = ((r_gemm, r_agg) if r_gemm['share_of_step'] >= r_agg['share_of_step']
                                        else (r_agg, r_gemm if gemms else None))
            roofline['note'] = note

    # the number a faster step moves: algorithmic bytes of the WHOLE step over the step time
    roofline_step = None
    if rank == 0:
        step_bytes = L * layer_algorithmic_bytes(stats[0], H, coboundary=coboundary, out_streams=3 if CINPP else 2)
        sgbs = step_bytes / (dt / args.steps) / 1e9
        roofline_step = {'bound': 'hbm', 'algorithmic_bytes_per_step': int(step_bytes), 'achieved': round(sgbs, 1),
                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(sgbs / HBM_PEAK_GBS, 4),
                         'frac_of_measured_achievable_6290': round(sgbs / 6290.0, 4),
                         'cells_per_s_at_6290': round(stats[0]['cells'] * L / (step_bytes / 6290e9), 1),
                         'note': 'SURVEY.md 8(d): gather-counted bytes of all layers of one step / ms_per_step'}

    mark('roofline')
    # ---- CPU baseline: the oracle on the host cores, bounded sample -----------------------------
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu:
        from oracle import cwn_oracle as O
        b = cpu_batches[0]
        feats = [[x.cpu() for x in f] for f in layer_inputs[0]]
        state = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        ocx = {'dimension': 2, 'y': None, 'cochains': [
            {k: b.cochains[d][k] for k in ('x', 'upper_index', 'lower_index', 'shared_boundaries',
                                           'shared_coboundaries', 'boundary_index', 'y', 'batch')}
            for d in range(3)]}

        def cpu_step():
            for l in range(L):
                for d in range(3):
                    ocx['cochains'][d]['x'] = feats[l][d]
                pre = f'convs.{l}.mp_levels.'
                for d, prm in enumerate(O.all_cochain_params(ocx, 2, include_down_features=False)):
                    if coboundary:
                        W, bias = state[f'{pre}{d}.msg_up_nn.1.weight'], state[f'{pre}{d}.msg_up_nn.1.bias']
                        msg = lambda xj, a: torch.relu(torch.cat([xj, a], -1) @ W.t() + bias)
                    else:
                        msg = lambda xj, a: xj
                    w = prm['x'].size(1)
                    O.propagate(prm['x'], prm['up_index'], None, prm['boundary_index'],
                                up_attr=prm['up_attr'], boundary_attr=prm['boundary_attr'],
                                message_up=msg, use_down_msg=False, up_msg_size=w, down_msg_size=w,
                                boundary_msg_size=w)
        # pick the thread count that is FASTEST for this (small-tensor) workload: all cores is
        # usually not it, and a baseline slowed down by oversubscription would flatter the GPU
        max_threads = torch.get_num_threads()
        trials = {}
        with torch.no_grad():
            for th in sorted({1, 4, 8, 16, 32, max_threads}):
                if th > max_threads:
                    continue
                torch.set_num_threads(th)
                cpu_step()
                k, t0 = 0, time.perf_counter()
                while time.perf_counter() - t0 < 0.5:
                    cpu_step()
                    k += 1
                trials[th] = k / (time.perf_counter() - t0)
            threads = max(trials, key=trials.get)
            torch.set_num_threads(threads)
            for _ in range(3):
                cpu_step()
            n, t0 = 0, time.perf_counter()
            while True:
                cpu_step()
                n += 1
                el = time.perf_counter() - t0
                if el > args.cpu_seconds or n >= 5000:
                    break
            torch.set_num_threads(max_threads)
        # secondary scope on the CPU: the oracle's whole EmbedSparseCIN forward, same batch
        cpu_full = None
        try:
            ocx_full = {'dimension': 2, 'y': None, 'num_complexes': b.num_complexes, 'cochains': [
                {k: b.cochains[d][k] for k in ('x', 'upper_index', 'lower_index', 'shared_boundaries',
                                               'shared_coboundaries', 'boundary_index', 'y', 'batch')}
                for d in range(3)]}
            for d in range(3):
                ocx_full['cochains'][d]['x'] = types[0][d]
            okw = {'zinc': dict(embed='zinc'), 'molhiv': dict(embed='ogb', readout='mean'),
                   'reddit': dict(embed=None, use_coboundaries=False, norm='id', jump_mode='cat')}[WL]
            with torch.no_grad():
                torch.set_num_threads(threads)
                O.sparse_cin_model_forward(state, ocx_full, L, **okw)
                k, t0 = 0, time.perf_counter()
                while time.perf_counter() - t0 < max(2.0, args.cpu_seconds / 5):
                    O.sparse_cin_model_forward(state, ocx_full, L, **okw)
                    k += 1
                cpu_full = stats[0]['cells'] * L * k / (time.perf_counter() - t0)
                torch.set_num_threads(max_threads)
        except Exception as e:   # the baseline leg must never take the bench down
            print(f'[bench] cpu full-forward baseline failed: {type(e).__name__}: {e}', file=sys.stderr)
        # ... and the oracle's forward (training-mode BatchNorm) + backward of the same model
        cpu_train = None
        try:
            st_g = {k: (v.clone().requires_grad_() if v.is_floating_point() and 'running' not in k
                        else v.clone()) for k, v in state.items()}

            def cpu_train_step():
                for v in st_g.values():
                    if v.requires_grad:
                        v.grad = None
                out_, _ = O.sparse_cin_model_forward(st_g, ocx_full, L, training=True, **okw)
                out_.abs().mean().backward()
            torch.set_num_threads(threads)
            cpu_train_step()
            k, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < max(2.0, args.cpu_seconds / 5):
                cpu_train_step()
                k += 1
            cpu_train = stats[0]['cells'] * L * k / (time.perf_counter() - t0)
            torch.set_num_threads(max_threads)
        except Exception as e:
            print(f'[bench] cpu train-step baseline failed: {type(e).__name__}: {e}', file=sys.stderr)
        cpu_value = stats[0]['cells'] * L * n / el
        try:
            with open('/proc/cpuinfo') as fh:
                cpu_model = next((ln.split(':', 1)[1].strip() for ln in fh if ln.startswith('model name')), 'unknown')
        except OSError:
            cpu_model = 'unknown'
        cpu_baseline = {'value': round(cpu_value, 1), 'unit': 'cells/s', 'cores': threads,
                        'kind': 'port', 'cpu_model': cpu_model,
                        'sample': f'{n} passes of the same propagate scope (12 propagate calls incl. '
                                  f'up_attr gathers) over batch 0 in {el:.1f} s, torch {torch.__version__} '
                                  f'CPU, {threads} threads (fastest of {sorted(trials)}; passes/s per thread count: '
                                  f'{ {k: round(v, 1) for k, v in trials.items()} }) of {os.cpu_count()} logical cores',
                        'gpu_over_cpu': round(value / cpu_value, 1),
                        'full_forward_cells_per_s': None if cpu_full is None else round(cpu_full, 1),
                        'train_fwd_bwd_cells_per_s': None if cpu_train is None else round(cpu_train, 1)}

    mark('cpu_baseline')
    # secondary: independent batches overlapped on the GPU (serving-style): S streams, each replaying
    # the step graph of its own batch; same kernels, same per-step work, K steps in total
    concurrent = None
    # (single-GPU leg: at N > 1 the whole-job number already comes from N processes side by side)
    if use_graph and not args.only_primary and len(batches) >= 2 and 'concurrent' not in SKIP and world == 1:
        try:
            S = min(4, len(batches))
            with torch.no_grad():
                streams_ = [torch.cuda.Stream() for _ in range(S)]
                cgraphs = []
                for si in range(S):
                    with torch.cuda.stream(streams_[si]):
                        propagate_scope(si)
                torch.cuda.synchronize()
                for si in range(S):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=streams_[si], capture_error_mode=CAPTURE_MODE):
                        keep = propagate_scope(si)
                    cgraphs.append((g, keep))
                rounds = max(args.steps // S, 5)

                def go(n):
                    for _ in range(n):
                        for si in range(S):
                            with torch.cuda.stream(streams_[si]):
                                cgraphs[si][0].replay()
                go(3)
                barrier()
                t0 = time.perf_counter()
                go(rounds)
                barrier()
                dtc = time.perf_counter() - t0
            if dist is not None:
                t = torch.tensor([dtc], device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dtc = float(t.item())
            ccells = torch.tensor([sum(stats[si]['cells'] for si in range(S)) * L * rounds], device=dev,
                                  dtype=torch.float64)
            if dist is not None:
                dist.all_reduce(ccells)
            concurrent = {'streams': S, 'cells_per_s': round(float(ccells.item()) / dtc, 1),
                          'steps': S * rounds, 'note': 'same step graphs, one per stream, replayed concurrently '
                          '(independent batches); the headline value is the sequential single-stream rate'}
        except Exception as e:
            print(f'[bench] concurrent-streams leg failed: {type(e).__name__}: {e}', file=sys.stderr)
            torch.cuda.synchronize()

    mark('concurrent')
    full_cells_total = float(full_cells.item())      # read here: result_line may run on the deadline thread, which must not touch the device

    def result_line(train, collate):
        """The ONE JSON line of rank 0 (everything but the last two legs is known before they start)."""
        s0 = stats[0]
        split_on = not ops.GEMM_EXACT
        dense = ('fp32 in / fp32 out; N = K = 128 products as exact 3-way bf16 operand splits on the bf16 MFMA pipe '
                 '(6 MFMAs per term, fp32 accumulate, max error 4e-7 of |x|.|w| vs float64 = the fp32-MFMA kernel\'s), '
                 'every other GEMM on fp32 MFMA; CWN_GEMM_SPLIT=0 runs all of them on fp32 MFMA'
                 if split_on and H == 128 else 'fp32 MFMA (v_mfma_f32_16x16x4_f32), exact fp32')
        out = {
            'metric': 'cells/sec, propagate scope, ' + {'zinc': 'ZINC-like ring-lifted batch (max_ring 6)', 'molhiv': 'molhiv-like ring-lifted batch (max_ring 6)', 'reddit': 'REDDIT-like clique-lifted batch (dim 2)'}[WL],
            'value': round(value, 1), 'unit': 'cells/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 5),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
            'data': 'synthetic', 'timing': timing,
            'config': {'workload': {'zinc': f'ZINC-like ring-lift (max_ring=6), {L}-layer ' + ('CIN++ (EmbedCINpp, mp/molec_models.py:167-199; three outputs per dimension)' if CINPP else 'SparseCIN') + f' propagate scope (hidden {H}, coboundary messages), batch {args.batch} per GPU [BASELINE configs[1]' + (' with CINppConv layers]' if CINPP else ']'), 'molhiv': f'ogbg-molhiv-like ring-lift (max_ring=6), {L}-layer OGBEmbedSparseCIN propagate scope (hidden {H}), batch {args.batch} per GPU [BASELINE configs[2]]', 'reddit': f'REDDIT-BINARY-like clique-lift (dim 2, hubs of degree >= 100), {L}-layer SparseCIN propagate scope (hidden {H}, no coboundaries, norm id, JK cat), batch {args.batch} per GPU [BASELINE configs[4]]'}[WL],
                       'key': f'{"zinc_cinpp" if CINPP else WL}:{args.batch}:{H}',
                       'batch_per_gpu': args.batch, 'hidden': H, 'layers': L,
                       'cells_per_batch': s0['cells'], 'N': [s0['N0'], s0['N1'], s0['N2']],
                       'E_up': [s0['E_up0'], s0['E_up1'], s0['E_up2']],
                       'B': [s0['B0'], s0['B1'], s0['B2']],
                       'launch': 'hipGraph replay' if use_graph else 'eager',
                       'plan_build_in_step': not BLOCKED, 'layer_kernel_form': FORM, 'layer_kernel': 'complex-blocked (1 launch per layer, COO in, no CSR plan)' if BLOCKED else 'grouped GEMM + CSR aggregation (2 launches per layer + 1 plan build per batch)', 'dense_arithmetic': dense,
                       'parallelism': f'replicas x{world} (no data-path collective)'},
            'roofline': roofline, 'roofline_other': roofline_other, 'roofline_mlp': roofline_mlp, 'roofline_plan_build': r_plan,
            'roofline_step': roofline_step,
            'cpu_baseline': cpu_baseline,
            'multi_gpu': MULTI,
            'secondary': {'full_forward_cells_per_s': (round(full_cells_total / dt_full, 1)
                                                       if dt_full == dt_full else None),     # leg skipped: null, not NaN
                          'full_forward_ms': round(dt_full / full_steps * 1e3, 5) if dt_full == dt_full else None,
                          'scope': 'EmbedSparseCIN forward: embedding, 4 conv layers incl. update '
                                   'MLPs + BatchNorm(eval), readout, head',
                          'forward_breakdown': forward_breakdown or None,
                          'collate': collate, 'concurrent_streams': concurrent, 'train_step': train,
                          'eager_launches': eager},
        }
        return out

    # N > 1: the legs from here on hold the job's only collectives over RCCL (the data-parallel training step, the
    # barrier at the end).  The headline number is complete at this point; should a collective never return
    # (nothing in this container can run more than one GPU), the line is printed without those legs and every
    # rank leaves, instead of the job dying in the RCCL watchdog with nothing on stdout.
    import threading
    printed, finished = threading.Event(), threading.Event()
    if world > 1:
        deadline = float(os.environ.get('CWN_BENCH_DP_DEADLINE_S', '240'))

        def _deadline():
            if finished.wait(deadline):
                return
            if rank == 0 and not printed.is_set():
                why = f'not finished {deadline:.0f} s after the single-GPU legs: skipped'
                emit(result_line({'skipped': why}, None))
            print(f'[bench] rank {rank}: data-parallel legs passed their deadline, leaving', file=sys.stderr, flush=True)
            os._exit(0)
        threading.Thread(target=_deadline, daemon=True).start()

    # secondary: the whole optimisation step (plans, forward, L1 loss, backward, fused Adam; for
    # N > 1 plus the ONE gradient all-reduce over RCCL), graph-captured -- SURVEY.md 8(d)/(e)
    train = None
    if not args.only_primary and 'train' not in SKIP:
        try:
            import copy
            from cwn_amd.train import TrainStep
            tmodel = copy.deepcopy(model).train()
            tb = [ComplexBatch.from_complex_list(gen(5000 + 1000 * rank + i), max_dim=2).to(dev)
                  for i in range(min(2, args.num_batches))]
            for b_ in tb:                       # targets of the prediction's shape where the
                if TASK == 'classification':        # synthetic generator has none
                    b_.y = torch.zeros(b_.num_complexes, dtype=torch.long, device=dev) if b_.y is None else b_.y.view(-1).long()
                elif b_.y is None or WL != 'zinc':
                    with torch.no_grad():
                        cs_ = [b_.cochains[d] for d in range(b_.dimension + 1)]
                        xs_keep = [c._x for c in cs_]
                        ref_pred = tmodel(b_)
                        for c, x_ in zip(cs_, xs_keep):
                            c._x = x_
                    b_.y = torch.zeros_like(ref_pred)
            trace = (lambda m: (torch.cuda.synchronize(), print(f'[bench trace r{rank}] {m}', file=sys.stderr, flush=True))) if os.environ.get('CWN_BENCH_TRACE') else (lambda m: None)
            trace('train: batches ready')
            # N > 1: eager by default.  The graph-captured data-parallel form (two graphs around the
            # all-reduce) is covered on one GPU by tests/test_gpu_parity.py, but it has never run over
            # RCCL on real multi-GPU hardware from this container, and a fault here would take the
            # scaling run's primary number down with it.  CWN_BENCH_TRAIN_GRAPH=1 opts in.
            train_graph = use_graph and (world == 1 or os.environ.get('CWN_BENCH_TRAIN_GRAPH') == '1')
            ts = TrainStep(tmodel, tb, task_type=TASK, use_graph=train_graph)
            trace('train: TrainStep built')
            tsteps = max(args.steps // 4, 10)
            for i in range(len(tb) + 2):
                ts.step(i % len(tb))
                trace(f'train: warm step {i}')
            barrier()
            t0 = time.perf_counter()
            for i in range(tsteps):
                ts.step(i % len(tb))
            barrier()
            dtt = time.perf_counter() - t0
            # the same steps, four behind one graph replay (TrainStep.steps): what a loop over a fixed epoch order pays
            dt4 = None
            if train_graph and world == 1:
                seq4 = [i % len(tb) for i in range(4)]
                ts.steps(seq4)
                ts.steps(seq4)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(max(tsteps // 4, 3)):
                    ts.steps(seq4)
                torch.cuda.synchronize()
                dt4 = (time.perf_counter() - t1) / (max(tsteps // 4, 3) * 4)
            if dist is not None:
                t = torch.tensor([dtt], device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dtt = float(t.item())
                # what the collective costs the step: the SAME steps with the bucket's reduce calls turned into no-ops (every
                # rank keeps its own gradient: numbers meaningless, kernels identical) -- exposed all-reduce time = the difference
                try:
                    bk = ts.bucket
                    saved = (bk.reduce_chunk, bk.all_reduce_mean, bk.finish)
                    bk.reduce_chunk = lambda *a, **k: None
                    bk.all_reduce_mean = lambda *a, **k: None
                    bk.finish = lambda *a, **k: None
                    for i in range(3):
                        ts.step(i % len(tb))
                    barrier()
                    t1 = time.perf_counter()
                    for i in range(tsteps):
                        ts.step(i % len(tb))
                    barrier()
                    dt_nc = time.perf_counter() - t1
                    bk.reduce_chunk, bk.all_reduce_mean, bk.finish = saved
                    t = torch.tensor([dt_nc], device=dev)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    dt_nc = float(t.item())
                    MULTI.update({'bucket_bytes': int(bk.flat.numel() * 4), 'bucket_chunks': int(ts.n_stages),
                                  'train_ms_per_step': round(dtt / tsteps * 1e3, 4),
                                  'train_ms_per_step_without_collectives': round(dt_nc / tsteps * 1e3, 4),
                                  'exposed_allreduce_ms_per_step': round((dtt - dt_nc) / tsteps * 1e3, 4),
                                  'allreduce_wire_floor_ms': round(2 * (world - 1) / world * bk.flat.numel() * 4 / 153e9 * 1e3, 4),
                                  'note': 'exposed = step with the chunked all-reduce issued inside the backward minus the same '
                                          'kernels with the collectives turned off (max over ranks); wire floor = ring all-reduce of '
                                          'the bucket at one xGMI link (~153 GB/s)'})
                except Exception as e:
                    MULTI['exposed_allreduce_error'] = f'{type(e).__name__}: {e}'
            tcells = torch.tensor([sum(batch_stats(tb[i % len(tb)])['cells'] for i in range(tsteps)) * L],
                                  device=dev, dtype=torch.float64)
            if dist is not None:
                dist.all_reduce(tcells)
            train = {'ms_per_step': round(dtt / tsteps * 1e3, 4),
                     'cells_per_s': round(float(tcells.item()) / dtt, 1), 'steps': tsteps,
                     'params': int(sum(p.numel() for p in ts.bucket.params)),
                     'backward_pieces': int(ts.n_stages), 'criterion': TASK, 'dropout_rate': DROP,
                     'ms_per_step_four_per_graph': None if dt4 is None else round(dt4 * 1e3, 4),
                     'scope': 'adjacency plans, weight packing, forward, L1 loss, backward (propagate steps: cwn_layer_bwd_own_f32; dense stages: '
                              'cwn_dense_stage_f32 / _bwd_f32), Adam on one flat buffer (cwn_adam_f32)'
                              + (f', {ts.bucket.flat.numel() * 4 / 1e6:.1f} MB flat gradient bucket all-reduced over RCCL in '
                                 f'{ts.n_stages} chunk(s), each issued as soon as the backward has left its layers'
                                 if world > 1 else '')
                              + ('; hipGraph replay' if train_graph else '; eager launches (host-bound)')}
            del ts, tmodel
        except Exception as e:
            print(f'[bench] train-step leg failed: {type(e).__name__}: {e}', file=sys.stderr)
            torch.cuda.synchronize()

    mark('train')
    # secondary: building the batch itself -- device-side collate from the HBM-resident packed
    # dataset vs the reference-style CPU collate (oracle restatement of data/complex.py:323-458)
    collate = None
    if rank == 0 and not args.only_primary and 'collate' not in SKIP:
        try:
            from cwn_amd.packed import PackedComplexes
            from cwn_amd.synthetic import zinc_like_complexes
            pool = [c for i in range(4) for c in gen(770 + i)]
            packed = PackedComplexes(pool, dev, max_dim=2)
            rng_idx = [torch.randperm(len(pool), generator=torch.Generator().manual_seed(i))[:args.batch].tolist()
                       for i in range(8)]
            for ix in rng_idx[:2]:
                packed.collate(ix)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(40):
                packed.collate(rng_idx[i % 8])
            torch.cuda.synchronize()
            dev_ms = (time.perf_counter() - t0) / 40 * 1e3
            collate = {'device_ms_per_batch': round(dev_ms, 4),
                       'scope': f'ComplexBatch of {args.batch} complexes from a packed HBM-resident dataset: host '
                                'segment tables + 1 H2D copy + 1 launch (cwn_collate)'}
            # the item table of the blocked layer kernel for such a batch (host: prefix sums -> cwn_layer_items_build
            # -> one H2D copy), rebuilt from scratch each time: what a new batch costs besides its collate
            if BLOCKED:
                nb_ = packed.collate(rng_idx[0])
                t0 = time.perf_counter()
                for i in range(20):
                    nb_._block_plan = None
                    nb_.block_plan().items(H, [True, True, False])
                torch.cuda.synchronize()
                table_ms = (time.perf_counter() - t0) / 20 * 1e3
                collate['item_table_host_ms_per_batch'] = round(table_ms, 4)
                # the headline excludes this host work.  Two figures that include it: the table of
                # EVERY step rebuilt from scratch in front of the step's graph replay (host and GPU overlap: the host
                # cuts the next table while the GPU runs), and the fully serial sum
                if use_graph:
                    with torch.no_grad():
                        gs = []
                        for bi in range(len(batches)):
                            g_ = torch.cuda.CUDAGraph()
                            with torch.cuda.graph(g_, capture_error_mode=CAPTURE_MODE):
                                keep_ = propagate_scope(bi)
                            gs.append((g_, keep_))
                        torch.cuda.synchronize()
                        n_ = 200
                        t0 = time.perf_counter()
                        for i in range(n_):
                            nb_._block_plan = None
                            nb_.block_plan().items(H, [True, True, False])
                            gs[i % len(gs)][0].replay()
                        torch.cuda.synchronize()
                        piped_ms = (time.perf_counter() - t0) / n_ * 1e3
                    cells_ = sum(stats[i % len(stats)]['cells'] for i in range(len(stats))) / len(stats) * L
                    step_ms = dt / args.steps * 1e3
                    collate['with_host_prep'] = {
                        'cells_per_s_overlapped': round(cells_ / (piped_ms * 1e-3), 1), 'ms_per_step_overlapped': round(piped_ms, 5),
                        'cells_per_s_serial': round(cells_ / ((step_ms + table_ms) * 1e-3), 1),
                        'ms_per_step_serial': round(step_ms + table_ms, 5),
                        'note': 'a fresh item table per step (host prefix sums -> cwn_layer_items_build -> H2D copy) + the step: '
                                'overlapped = table of step i+1 cut while the GPU runs step i (one host thread); serial = '
                                'the two times added'}
            if not args.no_cpu:
                from oracle import cwn_oracle as O
                keys = ('x', 'upper_index', 'lower_index', 'shared_boundaries', 'shared_coboundaries', 'boundary_index', 'y')
                odicts = [{'dimension': c.dimension, 'y': c.y, 'cochains': [
                    dict({k: c.cochains[d][k] for k in keys}, dim=d, num_cells=c.cochains[d].num_cells,
                         num_cells_up=c.cochains[d].num_cells_up, num_cells_down=c.cochains[d].num_cells_down,
                         batch=None) for d in range(c.dimension + 1)]} for c in pool]
                O.batch_complexes([odicts[i] for i in rng_idx[0]], max_dim=2)
                t0 = time.perf_counter()
                for i in range(10):
                    O.batch_complexes([odicts[j] for j in rng_idx[i % 8]], max_dim=2)
                collate['cpu_oracle_ms_per_batch'] = round((time.perf_counter() - t0) / 10 * 1e3, 3)
        except Exception as e:
            print(f'[bench] collate leg failed: {type(e).__name__}: {e}', file=sys.stderr)

    mark('collate')
    # secondary: the reference's loop as it really runs -- every step a batch it has never seen (data/data_loading.py:84-111
    # shuffles, exp/train_utils.py:35-75 steps through).  ONE captured graph per scope serves every batch of an epoch: the
    # collate, the segment tables, the item tables and every row count are device-side (cwn_amd/static_batch.py), the epoch's
    # permutation is uploaded once, a step is a graph replay and nothing else.
    fresh = None
    if rank == 0 and world == 1 and not args.only_primary and 'fresh' not in SKIP and use_graph and (BLOCKED or WL == 'reddit' or CINPP):
        try:
            fresh = fresh_batches_leg(args, model, gen, dev, H, L, rank, value / world, dt_full / full_steps * 1e3 if dt_full == dt_full else None,
                                      None if train is None else train.get('ms_per_step'), task=TASK,
                                      mode='blocked' if BLOCKED else 'csr')
            fresh['criterion'], fresh['dropout_rate'] = TASK, DROP
        except Exception as e:
            import traceback
            traceback.print_exc(file=sys.stderr)
            print(f'[bench] fresh-batches leg failed: {type(e).__name__}: {e}', file=sys.stderr)
            fresh = {'failed': f'{type(e).__name__}: {e}'}
            torch.cuda.synchronize()

    mark('fresh')
    # secondary: BASELINE configs[2] and configs[4] (molhiv-512, REDDIT-32) in the same default run -- value + roofline of
    # each from a `--brief` child process of this file (they existed only as builder-run files)
    workloads = None
    if (rank == 0 and world == 1 and WL == 'zinc' and not args.only_primary and 'workloads' not in SKIP
            and args.batch == 128 and H == 128):
        import subprocess
        workloads = {}
        # (+ the headline's own workload at batch 2048: the range where a launch has many items per CU and takes the
        # two-per-CU form of the layer kernel -- DESIGN.md 4.0)
        # (+ the headline's workload with the molecule sizes of the REAL ZINC subset -- 9 - 37 atoms, ~2 % beyond the 32 one
        # workgroup held at width 128 until round 4 (BIG records then; they fit since) -- at batch 128 and 2048: VERDICT r3 item 5)
        for wl in ('molhiv', 'molhiv_real_tail', 'reddit', 'zinc_cinpp', 'zinc_batch2048', 'zinc_real_spread', 'zinc_real_spread_batch2048'):
            try:
                extra = {'zinc_batch2048': ['--workload', 'zinc', '--batch', '2048', '--num-batches', '1'],
                         'zinc_real_spread': ['--workload', 'zinc'], 'zinc_cinpp': ['--workload', 'zinc'],
                         'molhiv_real_tail': ['--workload', 'molhiv'],
                         'zinc_real_spread_batch2048': ['--workload', 'zinc', '--batch', '2048', '--num-batches', '1']}.get(wl, ['--workload', wl])
                # (molhiv-512 = BASELINE configs[2] also runs its full forward, training step and the never-seen-batch legs)
                whole = wl in ('molhiv', 'molhiv_real_tail', 'reddit', 'zinc_cinpp')      # (reddit-32 = BASELINE configs[4]; zinc_cinpp: CIN++ layers)
                cmd = [sys.executable, os.path.abspath(__file__)] + extra + (['--no-cpu'] if whole else ['--brief']) + [
                       '--steps', str(max(args.steps, 20)),
                       '--warmup', str(max(args.warmup, 5)), '--kernel-reps', str(min(args.kernel_reps, 50))]
                env_ = dict(os.environ)
                if whole:
                    env_['CWN_BENCH_SKIP'] = 'eager,concurrent,collate,workloads'
                if wl.startswith('zinc_real_spread'):
                    env_['CWN_BENCH_ATOMS'] = 'zinc'
                if wl == 'molhiv_real_tail':
                    env_['CWN_BENCH_MOLHIV_TAIL'] = '5e-4'
                    env_['CWN_BENCH_ROUTED'] = '1'
                    env_['CWN_BENCH_SKIP'] = 'eager,concurrent,collate,workloads,roofline'
                if wl == 'zinc_cinpp':
                    env_['CWN_BENCH_MODEL'] = 'cinpp'
                import tempfile
                fd_, det_ = tempfile.mkstemp(prefix=f'cwn_bench_{wl}_', suffix='.json')
                os.close(fd_)
                env_['CWN_BENCH_DETAIL'] = det_
                t_wl = time.perf_counter()
                pr = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env_)
                LEGS['workload:' + wl] = time.perf_counter() - t_wl
                line = [ln for ln in pr.stdout.splitlines() if ln.startswith('{')]
                if pr.returncode != 0 or not line:
                    raise RuntimeError(f'rc {pr.returncode}: {pr.stderr[-300:]}')
                with open(det_) as fh_:
                    d_ = json.load(fh_)      # (the child's stdout line is the slim one; its detail file has the rest)
                os.unlink(det_)
                workloads[wl] = {'value': d_['value'], 'unit': d_['unit'], 'ms_per_step': d_['ms_per_step'],
                                 'workload': d_['config']['workload'], 'layer_kernel': d_['config']['layer_kernel'],
                                 'layer_kernel_form': d_['config'].get('layer_kernel_form'),
                                 'cells_per_batch': d_['config']['cells_per_batch'], 'timing': d_.get('timing'),
                                 'leg_seconds': d_.get('leg_seconds'),
                                 'roofline': {k: (d_['roofline'] or {}).get(k) for k in
                                              ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'avg_launch_us',
                                               'algorithmic_bytes_per_launch', 'traffic', 'frac_vs_pmc_traffic')},
                                 'roofline_step': {k: (d_['roofline_step'] or {}).get(k) for k in ('achieved', 'unit', 'frac')}}
                if whole:
                    sec_ = d_.get('secondary') or {}
                    fb_ = sec_.get('fresh_batches') or {}
                    workloads[wl].update({
                        'full_forward_ms': sec_.get('full_forward_ms'),
                        'train_step_ms': (sec_.get('train_step') or {}).get('ms_per_step'),
                        'fresh_batches': {k: fb_.get(k) for k in ('propagate', 'forward', 'train', 'every_batch_within_capacity',
                                                                   'device_error_word', 'steps_per_replay', 'batch', 'failed',
                                                                   'static_batch_mode', 'fill', 'routed')}})
            except Exception as e:
                workloads[wl] = {'failed': f'{type(e).__name__}: {e}'}
                print(f'[bench] workload {wl} failed: {type(e).__name__}: {e}', file=sys.stderr)

    mark('workloads')
    if rank == 0:
        printed.set()
        if dist is not None:
            # RCCL writes its version banner to the C stdout of rank 0, buffered until exit when stdout is a file: flush it NOW so
            # that the JSON line is the LAST line of stdout (the driver reads one line)
            try:
                import ctypes
                ctypes.CDLL(None).fflush(None)
            except Exception:
                pass
        line_ = result_line(train, collate)
        line_['secondary']['workloads'] = workloads
        line_['secondary']['fresh_batches'] = fresh
        line_['leg_seconds'] = {k: round(v, 1) for k, v in LEGS.items()}
        emit(line_)
    if dist is not None:
        dist.barrier()      # rank 0 runs the roofline / collate legs alone; leave together
        finished.set()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
