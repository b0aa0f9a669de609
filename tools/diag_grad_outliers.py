"""Diagnostic: the gradient tensors of the ZINC-128 training step where the product is further from float64 than the fp32 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import cwn_oracle as O
from tests._product import to_double
from tests.test_gpu_train_full import _oracle_cx, DEV
from cwn_amd.complex import ComplexBatch
from cwn_amd.models import EmbedSparseCIN
from cwn_amd.synthetic import zinc_like_complexes
from cwn_amd.train import TrainStep
torch.manual_seed(0)
L = 4
model = EmbedSparseCIN(28, 4, 1, L, 128, dropout_rate=0.0, max_dim=2, jump_mode=None, nonlinearity='relu', readout='sum', train_eps=False,
                       final_hidden_multiplier=2, final_readout='sum', init_reduce='sum', embed_edge=True, use_coboundaries=True, graph_norm='bn')
b = ComplexBatch.from_complex_list(zinc_like_complexes(128, 41, 6), max_dim=2)
model = model.to(DEV).train(); b = b.to(DEV)
state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
ocx = _oracle_cx(b)
def run(dtype):
    leaves = {k: v.to(dtype).clone().requires_grad_(True) for k, v in state.items() if v.is_floating_point() and 'running' not in k}
    st = dict(to_double(state)) if dtype == torch.float64 else dict(state)
    st.update(leaves)
    out, part = O.sparse_cin_model_forward(st, ocx, L, use_coboundaries=True, training=True, norm='bn', embed='zinc')
    y = b.y.detach().cpu().to(dtype).view(out.shape)
    (out - y).abs().mean().backward()
    return {k: v.grad for k, v in leaves.items()}, out.detach(), part
g64, o64, p64 = run(torch.float64)
g32, o32, _ = run(torch.float32)
ts = TrainStep(model, [b], task_type='regression', lr=1e-3, use_graph=os.environ.get('GRAPH', '1') == '1')
ts.step(0); torch.cuda.synchronize()
names = sys.argv[1:] or ['convs.3.mp_levels.1.update_boundaries_nn.3.weight', 'convs.3.mp_levels.1.update_boundaries_nn.0.weight',
                         'convs.3.mp_levels.1.update_up_nn.3.weight', 'convs.1.mp_levels.2.combine_nn.0.weight', 'convs.3.mp_levels.1.update_boundaries_nn.4.bias',
                         'convs.3.mp_levels.1.update_boundaries_nn.4.weight', 'convs.3.mp_levels.1.update_boundaries_nn.1.weight', 'convs.3.mp_levels.1.update_boundaries_nn.1.bias']
P = dict(model.named_parameters())
for n in names:
    g, r, r32 = P[n].grad.detach().cpu().double(), g64[n], g32[n].double()
    e = g - r
    line = f'{n}: |ref|max {float(r.abs().max()):.3e} |ref|rms {float(r.pow(2).mean().sqrt()):.3e}  err max {float(e.abs().max()):.3e} rms {float(e.pow(2).mean().sqrt()):.3e}  fp32 err max {float((r32 - r).abs().max()):.3e}'
    if e.dim() == 2:
        sv = torch.linalg.svdvals(e)
        line += f'  top singular share {float(sv[0] ** 2 / (sv ** 2).sum()):.3f}; err row-mean rms {float(e.mean(1).pow(2).mean().sqrt()):.2e} col-mean rms {float(e.mean(0).pow(2).mean().sqrt()):.2e}'
    print(line)
