"""L2 -> CU streaming rate when every workgroup reads the same buffer (tools/proto/l2_stream.hip)."""
import ctypes as C, os, torch
HERE = os.path.dirname(os.path.abspath(__file__))
L = C.CDLL(os.path.join(HERE, 'libproto_l2.so'))
L.proto_l2_stream.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
dev = torch.device('cuda:0')
out = torch.zeros(256 * 512 * 4, dtype=torch.int32, device=dev)
for kb in (96, 576, 2304):
    w = torch.randint(0, 2 ** 31 - 1, (kb * 256,), dtype=torch.int32, device=dev)
    for G in (32, 107, 214, 256):
        for mode in (0, 1):
            def go():
                assert L.proto_l2_stream(w.data_ptr(), kb * 1024, G, mode, out.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
            go(); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(20):
                    go()
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                g.replay()
            e1.record(); torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / 60
            print(f'{kb:5d} KB  G={G:3d} mode={mode}: {us:7.2f} us  {kb * 1024 * G / us / 1e6:6.2f} TB/s aggregate  '
                  f'{kb * 1024 / (us * 2100):5.1f} B/clk/CU (2.1 GHz)')
