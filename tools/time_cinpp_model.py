"""One training step of EmbedCINpp (mp/molec_models.py:167-199) at the ZINC shape -- hidden 128, 4 layers, batch 128, BatchNorm
in training mode, L1 loss, Adam -- replayed from TrainStep's captured graph: the update networks of the three streams on the
stage kernels (dense_train, a plan without combine stages) vs the same step with them as torch modules
(layers.FUSED_DENSE_TRAINING = False).   python tools/time_cinpp_model.py [batch] [hidden] [layers]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cwn_amd import layers
from cwn_amd.models import EmbedCINpp
from cwn_amd.synthetic import zinc_like_batch
from cwn_amd.train import TrainStep

dev = torch.device('cuda:0')
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 128
H = int(sys.argv[2]) if len(sys.argv) > 2 else 128
L = int(sys.argv[3]) if len(sys.argv) > 3 else 4


def measure(fused):
    layers.FUSED_DENSE_TRAINING = fused
    torch.manual_seed(0)
    model = EmbedCINpp(28, 4, 1, L, H, dropout_rate=0.0, max_dim=2, jump_mode=None, nonlinearity='relu', readout='sum',
                       train_eps=False, final_hidden_multiplier=2, final_readout='sum', init_reduce='sum', embed_edge=True,
                       use_coboundaries=True, graph_norm='bn').to(dev).train()
    batches = [zinc_like_batch(NB, seed=s, device=dev) for s in range(2)]
    ts = TrainStep(model, batches, task_type='regression', lr=1e-3, use_graph=True)
    for _ in range(3):
        for j in range(2):
            ts.step(j)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        for j in range(2):
            ts.step(j)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 40.0


try:
    a = measure(True)
    b = float('nan') if os.environ.get('CWN_ONLY_FUSED') == '1' else measure(False)      # (profiling: the fused step alone)
finally:
    layers.FUSED_DENSE_TRAINING = True
print('EmbedCINpp training step, batch %d, hidden %d, %d layers (replayed): %.3f ms with the update networks on the stage kernels, '
      '%.3f ms with them as torch modules' % (NB, H, L, a, b))
