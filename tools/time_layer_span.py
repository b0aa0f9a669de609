"""Where a launch of the complex-blocked layer kernel spends its time OUTSIDE its workgroups' own chains:
every wave stamps its first and last instruction on the chip-wide 100-MHz clock (s_memrealtime, the timing build:
    make -C cwn_amd/csrc timing && CWN_HIP_LIB=$PWD/cwn_amd/libcwn_hip_timing.so python tools/time_layer_span.py [batch] [F]
), N back-to-back launches of the headline's layer (ZINC-like batch, CSR-load mode, the <F, 2> instantiation) are captured in
one hipGraph, and the stamps of the last replay give, per launch:
    period            first wave of launch i + 1  -  first wave of launch i          (= what bench.py times per launch)
    boundary          first wave of launch i + 1  -  last wave end of launch i       (drain + dispatch of the next grid)
    ramp              last workgroup's first wave -  first workgroup's first wave    (how long the dispatcher takes to place 256 x 16 waves)
    wave skew         within a workgroup: last wave start - first wave start
    chain             a workgroup's last wave end - its first wave start             (distribution over workgroups)
    critical path     the workgroup that ends last: when it started, how long it ran
The resolution of the clock is 10 ns.  Prints a markdown report (profiles/r4_layer_span.md)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cwn_amd import _ffi, ops                               # noqa: E402
from cwn_amd.complex import ComplexBatch                    # noqa: E402
from cwn_amd.layers import SparseCINConv                    # noqa: E402
from cwn_amd.synthetic import zinc_like_complexes           # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
F = int(sys.argv[2]) if len(sys.argv) > 2 else 128
N_LAUNCH = 8
REC = 96
dev = torch.device('cuda:0')
L = _ffi.lib()
assert hasattr(L, 'cwn_layer_debug_stamps_many'), 'needs the timing build (CWN_HIP_LIB=.../libcwn_hip_timing.so)'
L.cwn_layer_debug_stamps_many.argtypes = [C.c_void_p, C.c_longlong]
L.cwn_layer_debug_stamps_many.restype = None
b = ComplexBatch.from_complex_list(zinc_like_complexes(n, 0, 6), max_dim=2).to(dev)
for d in range(3):
    b.cochains[d].x = torch.randn(b.cochains[d].num_cells, F, device=dev)
torch.manual_seed(0)
conv = SparseCINConv(F, F, F, None, None, None, None, max_dim=2, hidden=F, act_module=torch.nn.ReLU,
                     layer_dim=F, use_coboundaries=True).to(dev).eval()
params = b.get_all_cochain_params(max_dim=2, include_down_features=False)
with torch.no_grad():
    args = conv._blocked_args(params, 0)
assert not isinstance(args, str), args
dims, plan, table, key = args
n_items = int(table.items.size(0))
launch = ops.LayerLaunch(dims, table)
xs = [D.x for D in dims]
stamps = torch.zeros(N_LAUNCH, n_items, REC, dtype=torch.int64, device=dev)
with torch.no_grad():
    launch.run(xs, _ffi.LAYER_CSR_STORE)
    for _ in range(3):
        launch.run(xs, _ffi.LAYER_CSR_LOAD)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        launch.run(xs, _ffi.LAYER_CSR_LOAD)
    torch.cuda.current_stream().wait_stream(side)
    L.cwn_layer_debug_stamps_many(stamps.data_ptr(), N_LAUNCH)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(N_LAUNCH):
            launch.run(xs, _ffi.LAYER_CSR_LOAD)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    per_launch_us = e0.elapsed_time(e1) * 1e3 / (20 * N_LAUNCH)
L.cwn_layer_debug_stamps_many(None, 1)
st = stamps.cpu().numpy().astype(np.int64)                  # [launch][item][96]
TICK = 0.01                                                 # us per tick of the 100-MHz clock
start, end = st[:, :, 64:80], st[:, :, 80:96]               # per wave
live = (start > 0) & (end > 0)
wg_live = live.any(axis=2)
big = np.iinfo(np.int64).max
s_wg = np.where(live, start, big).min(axis=2)               # a workgroup's first wave
s_last = np.where(live, start, 0).max(axis=2)               # ... its last wave to start
e_wg = np.where(live, end, 0).max(axis=2)                   # ... its last wave to end
print(f'# layer_kernel<{F}, load> at the ZINC-like batch of {n}: {n_items} workgroups of 16 waves, {N_LAUNCH} back-to-back launches in one hipGraph')
print(f'\nHIP events around 20 replays: **{per_launch_us:.2f} us per launch** (timing build: the stamps cost ~0.2 us)\n')
print('| launch | period us | in-kernel span us | boundary to next us | ramp (last wg start - first) us | wave skew in a wg (mean / max) us | chain per wg mean / p95 / max us | the workgroup that ends last: started at / ran for us |')
print('|---|---|---|---|---|---|---|---|')
rows = []
for i in range(N_LAUNCH):
    m = wg_live[i]
    S, E = s_wg[i][m].min(), e_wg[i][m].max()
    nxt = s_wg[i + 1][wg_live[i + 1]].min() if i + 1 < N_LAUNCH else None
    chain = (e_wg[i][m] - s_wg[i][m]) * TICK
    skew = (s_last[i][m] - s_wg[i][m]) * TICK
    last = np.argmax(np.where(m, e_wg[i], 0))
    rows.append(((nxt - S) * TICK if nxt else float('nan'), (E - S) * TICK, (nxt - E) * TICK if nxt else float('nan'),
                 (s_wg[i][m].max() - S) * TICK, skew.mean(), skew.max(), chain.mean(), np.percentile(chain, 95), chain.max(),
                 (s_wg[i][last] - S) * TICK, (e_wg[i][last] - s_wg[i][last]) * TICK))
    r = rows[-1]
    print(f'| {i} | {r[0]:.2f} | {r[1]:.2f} | {r[2]:.2f} | {r[3]:.2f} | {r[4]:.2f} / {r[5]:.2f} | {r[6]:.2f} / {r[7]:.2f} / {r[8]:.2f} | {r[9]:.2f} / {r[10]:.2f} |')
a = np.array(rows[1:-1])                                    # (the first launch of a replay follows the graph launch itself)
print(f'\nmean over launches 1 .. {N_LAUNCH - 2}: period {a[:, 0].mean():.2f} us = in-kernel span {a[:, 1].mean():.2f} + boundary {a[:, 2].mean():.2f}; '
      f'ramp {a[:, 3].mean():.2f}; chain mean {a[:, 6].mean():.2f} / max {a[:, 8].mean():.2f}; the last workgroup to end started {a[:, 9].mean():.2f} us '
      f'after the first and ran {a[:, 10].mean():.2f} us')
# start order: does the dispatcher place workgroups in blockIdx order, and how fast?
i = N_LAUNCH // 2
order = np.argsort(np.where(wg_live[i], s_wg[i], big))
rel = (s_wg[i][order] - s_wg[i][order[0]]) * TICK
q = [0, 31, 63, 127, 191, min(255, len(order) - 1)]
print(f'\nlaunch {i}: start time of the k-th workgroup to start (us after the first): ' + ', '.join(f'k={k}: {rel[k]:.2f}' for k in q if k < len(rel)))

# ---- round 6: WHICH workgroups are the slow ones?  chain length against what the item holds -----------
it = table.items.cpu().numpy().astype(np.int64)             # [n_items][32] records (include/cwn_hip.h)
i = N_LAUNCH // 2
chain_us = (e_wg[i] - s_wg[i]) * TICK
g_dim, n_g, n_c, e_up = it[:, 1], it[:, 3], it[:, 5], it[:, 7]
b0, b1 = it[:, 9 + 4], it[:, 16 + 4]
n_t1 = it[:, 16 + 2]
tiles = -(-n_g // 16) + -(-n_c // 16)
rounds = np.maximum(-(-n_g // 32), -(-n_t1 // 32))            # rounds of the reduce phases (32 lane groups at F = 128)
print(f'\n## launch {i}: chain of a workgroup against what its item holds\n')
print('| set | items | chain mean / max us | cells g (mean / max) | row tiles (mean / max) | entries up + boundary (mean / max) |')
print('|---|---|---|---|---|---|')
for g in sorted(set(g_dim.tolist())):
    m = (g_dim == g) & wg_live[i]
    ent = (e_up + b0 + b1)[m]
    print(f'| g = {g} | {int(m.sum())} | {chain_us[m].mean():.2f} / {chain_us[m].max():.2f} | {n_g[m].mean():.1f} / {n_g[m].max()} | '
          f'{tiles[m].mean():.2f} / {tiles[m].max()} | {ent.mean():.0f} / {ent.max()} |')
print('\nchain by number of 16-row tiles the item multiplies (all sets):')
for t in sorted(set(tiles[wg_live[i]].tolist())):
    m = (tiles == t) & wg_live[i]
    print(f'  {t} tiles: {int(m.sum()):3d} items, chain mean {chain_us[m].mean():.2f} us, max {chain_us[m].max():.2f}')
print('\nchain by rounds of the reduce phases (cells of a task beyond 32 take a second round):')
for r in sorted(set(rounds[wg_live[i]].tolist())):
    m = (rounds == r) & wg_live[i]
    print(f'  {r} round(s): {int(m.sum()):3d} items, chain mean {chain_us[m].mean():.2f} us, max {chain_us[m].max():.2f}')
order = np.argsort(-np.where(wg_live[i], chain_us, 0))[:12]
print('\nthe twelve longest chains: workgroup | set | chain us | cells g | cells g+1 | tiles | up entries | boundary entries (task 0 + 1) | started at us')
S0 = s_wg[i][wg_live[i]].min()
for w in order:
    print(f'  {int(w):4d} | g = {int(g_dim[w])} | {chain_us[w]:.2f} | {int(n_g[w])} | {int(n_c[w])} | {int(tiles[w])} | {int(e_up[w])} | {int(b0[w])} + {int(b1[w])} | {(s_wg[i][w] - S0) * TICK:.2f}')
cc = np.corrcoef(np.stack([chain_us[wg_live[i]], tiles[wg_live[i]], (e_up + b0 + b1)[wg_live[i]], n_g[wg_live[i]]]))[0]
print(f'\ncorrelation of the chain with: row tiles {cc[1]:.2f}, entries {cc[2]:.2f}, cells of g {cc[3]:.2f}')
