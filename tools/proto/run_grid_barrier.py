"""Cost of a grid-wide barrier inside one launch (tools/proto/grid_barrier.hip), graph-replayed:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared tools/proto/grid_barrier.hip -o tools/proto/libproto_bar.so
    python tools/proto/run_grid_barrier.py"""
import ctypes as C
import os
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
L = C.CDLL(os.path.join(HERE, 'libproto_bar.so'))
L.proto_grid_barrier.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint, C.c_void_p, C.c_void_p]
dev = torch.device('cuda:0')
x = torch.randn(256 * 4096, device=dev)
y = torch.zeros(256 * 4096, device=dev)
err = torch.zeros(1, dtype=torch.int32, device=dev)
for G in (64, 214, 256):
    for nb in (0, 1, 3, 6):
        counters = torch.zeros(16, dtype=torch.int32, device=dev)
        epoch = [0]
        def go():
            epoch[0] += 1
            assert L.proto_grid_barrier(x.data_ptr(), y.data_ptr(), counters.data_ptr(), G, nb, epoch[0], err.data_ptr(),
                                        torch.cuda.current_stream().cuda_stream) == 0
        go(); torch.cuda.synchronize()
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            go()
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        reps = 50
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            counters.zero_()              # the graph restarts the epochs on every replay
            epoch[0] = 0
            for _ in range(reps):
                go()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        print(f'G={G:4d} barriers={nb}: {1e3 * e0.elapsed_time(e1) / (4 * reps):7.2f} us/launch   err={int(err.item())}')
