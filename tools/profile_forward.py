"""Run the EmbedSparseCIN forward (eval, no_grad) a few times, eagerly, for rocprofv3."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cwn_amd.models import EmbedSparseCIN
from cwn_amd.synthetic import zinc_like_batch

dev = torch.device('cuda:0')
torch.manual_seed(0)
model = EmbedSparseCIN(28, 4, 1, 4, 128, dropout_rate=0.0, embed_edge=True, use_coboundaries=True).to(dev).eval()
b = zinc_like_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 128, seed=0, device=dev)
vt, et = b.cochains[0].x.clone(), b.cochains[1].x.clone()
with torch.no_grad():
    for i in range(6):
        b.cochains[0]._x, b.cochains[1]._x, b.cochains[2]._x = vt, et, None
        if i == 5:
            torch.cuda.synchronize()
            print('MARK')
        y = model(b)
    torch.cuda.synchronize()
print(float(y.sum()))
