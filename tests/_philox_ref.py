"""numpy restatement of Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) and of
the keep rule of include/cwn_hip.h: cwn_dropout -- the CHECKER of csrc/cwn_dropout.h.  Test infrastructure only."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Arrays (or scalars) of uint32 counters and keys -> four uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(v, dtype=np.uint32) for v in (c0, c1, c2, c3))
    k0, k1 = np.uint32(k0), np.uint32(k1)
    with np.errstate(over='ignore'):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & MASK).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & MASK).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0, k1 = np.uint32(k0 + W0), np.uint32(k1 + W1)
    return c0, c1, c2, c3


def multipliers(shape, p, seed, step, site):
    """What cwn_dropout multiplies the elements of a matrix of `shape` with (element e = flat row-major index)."""
    n = int(np.prod(shape))
    q = np.arange((n + 3) // 4, dtype=np.uint32)
    r = philox4x32_10(q, np.full_like(q, site), np.full_like(q, step & 0xFFFFFFFF), np.full_like(q, (step >> 32) & 0xFFFFFFFF),
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    r = np.stack(r, axis=1).reshape(-1)[:n]
    p32 = np.float32(p)
    t = float(p32) * 4294967296.0
    thresh = np.uint32(0xFFFFFFFF) if t >= 4294967295.0 else np.uint32(int(t))
    scale = np.float32(1.0) / (np.float32(1.0) - p32)
    return np.where(r >= thresh, scale, np.float32(0.0)).astype(np.float32).reshape(shape)
