"""Build-time facts about the kernels of the built library, read from the code objects' metadata (no GPU): the
kernels of the headline path keep their registers (a spill there is a scratch round trip behind a full
`s_waitcnt vmcnt(0)` -- docs/history/DESIGN_rounds1-5.md 4.0 lost 4 us per step to four spilled registers once), fit the occupancy they
were written for, and no kernel starts spilling unnoticed."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from kernel_resources import kernels, short, READELF      # noqa: E402

LIB = os.path.join(ROOT, 'cwn_amd', 'libcwn_hip.so')

# kernels of the CSR path that spill today (round-1 code; candidates for the next measurement)
KNOWN_SPILLS = {'gemm_kernel<Lb0ELb0ELi128ELi2ELb0ELi2ELb0E>', 'gemm_kernel<Lb0ELb1ELi128ELi2ELb0ELi2ELb0E>',
                'gemm_kernel<Lb1ELb0ELi128ELi2ELb1ELi2ELb0E>',
                # the transposed-weight kernel with the BatchNorm-backward prologue: a second staged tile and 20 per-column
                # constants on top of the stationary weight fragments (26 registers in scratch, outside the MFMA loop)
                'gemm_kernel<Lb1ELb0ELi128ELi4ELb1ELi2ELb1E>',
                'gemm_kernel<Lb1ELb0ELi128ELi2ELb1ELi2ELb1E>',        # the same at 64 x 64 tiles (K = 128, N <= 64: no model of the reference has it)
                'aggregate_kernel<Li4ELb0ELb1E>', 'aggregate_kernel<Li4ELb1ELb1E>'}


@pytest.fixture(scope='module')
def table():
    if not (os.path.exists(LIB) and os.path.exists(READELF)):
        pytest.skip('library or llvm-readelf missing')
    ks = {short(n): v for n, v in kernels(LIB).items()}
    assert len(ks) > 60, len(ks)                     # every .hip file contributed its code object
    return ks


def test_headline_kernels_keep_their_registers(table):
    layer = {n: v for n, v in table.items() if n.startswith('layer_kernel<')}
    # F in {64, 128} x {sort, sort + store, load}, in two forms (cwn_layer.hip compiled twice): 16 waves, one workgroup
    # per CU; 8 waves within 128 registers, two per CU
    assert len(layer) == 12 and sorted(v['max_flat_workgroup_size'] for v in layer.values()) == [512] * 6 + [1024] * 6
    for n, v in layer.items():
        # 1024 threads = 16 waves = four per SIMD: 512 / 4 = 128 registers a lane; 2 x 512 threads: the same; none in scratch
        assert v['vgpr_count'] <= 128, (n, v)
        assert v['vgpr_spill_count'] == v['sgpr_spill_count'] == v['private_segment_fixed_size'] == 0, (n, v)
    mlp = {n: v for n, v in table.items() if n.startswith('update_mlp_kernel<')}
    assert len(mlp) >= 3              # <128>, <64>: alternating, one workgroup per CU; sequential (round 5): two per CU
    for n, v in mlp.items():
        # 512 threads = two waves per SIMD; the sequential schedule shares its CU with a second workgroup: four per SIMD
        two_per_cu = 'Li2ELb1E' in n               # update_mlp_kernel<F, RT = 2, SEQ = true, NARROW>
        assert v['max_flat_workgroup_size'] == 512 and v['vgpr_count'] <= (128 if two_per_cu else 256), (n, v)
        assert v['vgpr_spill_count'] == v['sgpr_spill_count'] == v['private_segment_fixed_size'] == 0, (n, v)
    assert any('Li2ELb1E' in n for n in mlp) and len(mlp) == 8
    for n, v in table.items():
        if n.startswith(('gemm_split_kernel<', 'gemm_tn_kernel<', 'norm_kernel<', 'gather_rows_kernel<')) or 'collate_kernel' in n:
            assert v['vgpr_spill_count'] == 0 and v['private_segment_fixed_size'] == 0, (n, v)


def test_no_kernel_starts_spilling_unnoticed(table):
    spilling = {n for n, v in table.items() if v['vgpr_spill_count'] or v['private_segment_fixed_size']}
    assert spilling <= KNOWN_SPILLS, sorted(spilling - KNOWN_SPILLS)
