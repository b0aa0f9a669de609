"""Per-launch cost on this stack: N trivial C-ABI launches (gather of one row) replayed from a
hipGraph vs issued eagerly through ctypes, timed with events and wall clock."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cwn_amd import _ffi

dev = torch.device('cuda:0')
L = _ffi.lib()
src = torch.randn(64, 128, device=dev)
idx = torch.zeros(1, dtype=torch.long, device=dev)
out = torch.empty(1, 128, device=dev)
N = 200
args = (src.data_ptr(), 64, 128, idx.data_ptr(), 1, out.data_ptr())


def launch_n(stream):
    for _ in range(N):
        L.cwn_gather_rows_f32(*args, stream)


s = torch.cuda.current_stream().cuda_stream
launch_n(s); torch.cuda.synchronize()
# eager
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record(); launch_n(s); e1.record(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f'eager : host issue {1e6 * (t1 - t0) / N:.2f} us/launch, gpu span {1e3 * e0.elapsed_time(e1) / N:.2f} us/launch, wall {1e6 * (t2 - t0) / N:.2f}')
# graph
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    launch_n(torch.cuda.current_stream().cuda_stream)
g.replay(); torch.cuda.synchronize()
t0 = time.perf_counter(); e0.record()
for _ in range(5):
    g.replay()
e1.record(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f'graph : gpu span {1e3 * e0.elapsed_time(e1) / (5 * N):.2f} us/launch, wall {1e6 * (t2 - t0) / (5 * N):.2f}')
