"""Kernel-level timing of the plan build (cwn_csr_build) for one workload's batch, eager launches:
   rocprofv3 --kernel-trace --stats -- python tools/profile_plan_build.py reddit"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cwn_amd import csr, synthetic


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else 'reddit'
    dev = torch.device('cuda:0')
    if wl == 'reddit':
        from cwn_amd.complex import ComplexBatch
        cx = synthetic.reddit_like_complexes(32, seed=0)
        b = ComplexBatch.from_complex_list(cx, max_dim=2).to(dev)
    else:
        b = synthetic.zinc_like_batch(int(sys.argv[2]) if len(sys.argv) > 2 else 128, seed=0).to(dev)
    for _ in range(30):
        csr._cache.clear()
        b.prepare(max_dim=2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        csr._cache.clear()
        b.prepare(max_dim=2)
    e1.record()
    torch.cuda.synchronize()
    print(f'{wl}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per plan build (eager, host included)')


if __name__ == '__main__':
    main()
