"""cwn_amd -- MI355X-native cellular message passing (the hot path of twitter-research/cwn).

    cell_mp   CochainMessagePassing (drop-in propagate), CochainMessagePassingParams, IndexedRows
    layers    SparseCINConv / CINConv / ... built on it (mirror of the reference's mp/layers.py)
    complex   Cochain / Complex / ComplexBatch (input format, mirror of data/complex.py)
    csr       per-batch int32 CSR plans        ops   differentiable ops over the C ABI
    _ffi      ctypes binding of libcwn_hip.so (include/cwn_hip.h); csrc/ holds the HIP kernels

GPU only: the product path has no CPU fallback (oracle/ is the CPU checker, used by tests only).
"""
__version__ = '0.1.0'
