"""Host time of the eager full forward (EmbedSparseCIN, ZINC-128, inference): total and a cProfile by own time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cwn_amd.models import EmbedSparseCIN
from cwn_amd.complex import ComplexBatch
from cwn_amd.synthetic import zinc_like_complexes
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = EmbedSparseCIN(28, 4, 1, 4, 128, dropout_rate=0.0, embed_edge=True, use_coboundaries=True).to(dev).eval()
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 1          # distinct batches cycled (bench.py cycles its --num-batches)
bs = [ComplexBatch.from_complex_list(zinc_like_complexes(128, i, 6), max_dim=2).to(dev) for i in range(NB)]
x0s = [[None if b.cochains[d].x is None else b.cochains[d].x.clone() for d in range(3)] for b in bs]
b = bs[0]
_i = [0]
def fwd():
    k = _i[0] % NB
    _i[0] += 1
    bb = bs[k]
    for d in range(3):        # (the model overwrites the batch's features layer by layer, as the reference does)
        bb.cochains[d]._x = x0s[k][d]
    return model(bb)
with torch.no_grad():
    for _ in range(20): fwd()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300): fwd()
    torch.cuda.synchronize()
    print(f'eager model(batch): {(time.perf_counter() - t0) / 300 * 1e6:.0f} us')
    # real host time per segment (perf_counter around the model's pieces; cProfile inflates everything ~2 x)
    T = {}
    def timed(obj, name, label):
        fn = getattr(obj, name)
        def wrapped(*a, **k):
            t = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                T[label] = T.get(label, 0.0) + time.perf_counter() - t
        setattr(obj, name, wrapped)
    timed(model.init_conv, 'forward', 'front (init_conv)')
    timed(model, '_head_fused', 'head (_head_fused)')
    for conv in model.convs:
        timed(conv, 'propagate_all', 'conv: propagate_all')
        timed(conv, '_dense_eval', 'conv: _dense_eval')
        timed(conv, 'forward', 'conv: forward (incl. the two above)')
    for bb in bs:
        timed(bb, 'get_all_cochain_params', 'batch.get_all_cochain_params (5 calls)')
        timed(bb, 'set_xs', 'batch.set_xs (5 calls)')
    from cwn_amd import csr
    real_check = csr.check_errors
    def check(dev):
        t = time.perf_counter(); real_check(dev); T['error word (sync)'] = T.get('error word (sync)', 0.0) + time.perf_counter() - t
    csr.check_errors = check
    for _ in range(20): fwd()
    torch.cuda.synchronize(); T.clear()
    t0 = time.perf_counter()
    for _ in range(300): fwd()
    torch.cuda.synchronize()
    print(f'with segment timers: {(time.perf_counter() - t0) / 300 * 1e6:.0f} us')
    for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
        print(f'  {v / 300 * 1e6:7.1f} us  {k}')
    if os.environ.get('CPROFILE'):
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable()
        for _ in range(100): fwd()
        pr.disable()
        pstats.Stats(pr).sort_stats('tottime').print_stats(22)
        pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
