"""Register, scratch and LDS use of every kernel in the BUILT library, read from the code objects' metadata
(no GPU needed):  python tools/kernel_resources.py [path/to/libcwn_hip.so]

The gfx950 code objects are the AMDGPU ELFs embedded in the library's .hip_fatbin section; llvm-readelf --notes
prints their per-kernel metadata.  tests/test_kernel_resources.py pins the kernels of the headline path on it
(a spill there costs a scratch round trip behind `s_waitcnt vmcnt(0)`: DESIGN.md 4.0)."""
import os
import re
import struct
import subprocess
import sys
import tempfile

READELF = '/opt/rocm/lib/llvm/bin/llvm-readelf'
FIELDS = ('vgpr_count', 'agpr_count', 'sgpr_count', 'vgpr_spill_count', 'sgpr_spill_count',
          'private_segment_fixed_size', 'group_segment_fixed_size', 'max_flat_workgroup_size')


def kernels(lib_path: str) -> dict:
    """{mangled kernel name: {field: int}} over every gfx950 code object of the library."""
    blob = open(lib_path, 'rb').read()
    out, pos = {}, 0
    while True:
        i = blob.find(b'\x7fELF', pos)
        if i < 0:
            return out
        pos = i + 4
        if blob[i + 4] != 2 or struct.unpack_from('<H', blob, i + 18)[0] != 224:      # ELF64, EM_AMDGPU
            continue
        e_shoff = struct.unpack_from('<Q', blob, i + 40)[0]
        e_shentsize, e_shnum = struct.unpack_from('<HH', blob, i + 58)
        with tempfile.NamedTemporaryFile(suffix='.elf', delete=False) as f:
            f.write(blob[i:i + e_shoff + e_shentsize * e_shnum])
        try:
            notes = subprocess.run([READELF, '--notes', f.name], capture_output=True, text=True, check=True).stdout
        finally:
            os.unlink(f.name)
        for blk in re.split(r'\n\s+- \.agpr_count', notes)[1:]:
            blk = '.agpr_count' + blk
            name = re.search(r'\.name:\s+(\S+)', blk).group(1)
            v = {k: int(re.search(r'\.' + k + r':\s+(\d+)', blk).group(1)) for k in FIELDS}
            # the same source compiled twice (cwn_layer.hip: the 16-wave and the two-per-CU form) gives two code objects
            # with the same kernel names: keep both, the second one under name@<threads>
            if name in out and out[name] != v:
                name = f'{name}@{v["max_flat_workgroup_size"]}'
            out[name] = v


def short(name: str) -> str:
    name, _, tag = name.partition('@')
    m = re.match(r'_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)I(.*)EEv', name)
    return (f'{m.group(1)}<{m.group(2)}>' if m else name) + (f'@{tag}' if tag else '')


if __name__ == '__main__':
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ks = kernels(sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, 'cwn_amd', 'libcwn_hip.so'))
    print(f'{"kernel":58s} vgpr sgpr spill(v/s) scratch  threads')
    for n, v in sorted(ks.items()):
        print(f'{short(n)[:58]:58s} {v["vgpr_count"]:4d} {v["sgpr_count"]:4d} {v["vgpr_spill_count"]:5d}/{v["sgpr_spill_count"]:<3d} '
              f'{v["private_segment_fixed_size"]:7d} {v["max_flat_workgroup_size"]:8d}')
