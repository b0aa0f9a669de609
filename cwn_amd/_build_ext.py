"""Builds cwn_amd/_cwn_torch_ext.so, the compiled binding of the eager path (csrc/cwn_torch_ext.cpp): host C++ against the
torch headers, compiled with plain g++ (no device code, no hipify pass), in-tree so that the built module travels with the
repository snapshot.  `python -m cwn_amd._build_ext` or `__graft_entry__.build()`."""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'csrc', 'cwn_torch_ext.cpp')
NAME = '_cwn_torch_ext'
OUT = os.path.join(HERE, NAME + '.so')


def up_to_date() -> bool:
    deps = [SRC, os.path.join(HERE, '..', 'include', 'cwn_hip.h'), os.path.abspath(__file__)]
    return os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if up_to_date() and not force:
        return OUT
    import torch
    from torch.utils import cpp_extension as ce
    rocm = os.environ.get('ROCM_PATH', '/opt/rocm')
    cxx = os.environ.get('CXX', 'g++')
    cmd = [cxx, '-O2', '-std=c++17', '-fPIC', '-shared', '-Wall', '-Wno-unused-function',
           f'-DTORCH_EXTENSION_NAME={NAME}', '-DTORCH_API_INCLUDE_EXTENSION_H', '-D__HIP_PLATFORM_AMD__=1', '-DUSE_ROCM=1',
           f'-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}']
    cmd += [f'-I{p}' for p in ce.include_paths()] + [f'-I{rocm}/include', f'-I{sysconfig.get_paths()["include"]}']
    cmd += [SRC, '-o', OUT]
    for lp in ce.library_paths():
        cmd += [f'-L{lp}', f'-Wl,-rpath,{lp}']
    cmd += ['-lc10', '-lc10_hip', '-ltorch_cpu', '-ltorch_hip', '-ltorch', '-ltorch_python']
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(OUT)
