"""The compiled binding of the eager path (csrc/cwn_torch_ext.cpp -> _cwn_torch_ext.so, built by _build_ext.py): the per-call
part of a prepared launch in C++.  `ext()` returns the module or None -- None when it was not built (ctypes then does
everything, as before round 5), when CWN_BINDING=ctypes, or inside `binding('ctypes')`."""
import contextlib
import os

from . import _ffi

_mod = None
_state = {'tried': False, 'forced': os.environ.get('CWN_BINDING') or None}


def ext():
    if _state['forced'] == 'ctypes':
        return None
    if not _state['tried']:
        _state['tried'] = True
        global _mod
        try:
            from . import _cwn_torch_ext as m
        except ImportError:
            if _state['forced'] == 'compiled':
                raise
            m = None
        if m is not None and int(m.abi_version) != _ffi.ABI_VERSION:
            raise _ffi.CwnError(f'_cwn_torch_ext.so was built against ABI {int(m.abi_version)}, the package speaks '
                                f'{_ffi.ABI_VERSION}: python -m cwn_amd._build_ext --force')
        _mod = m
    return _mod


def active() -> str:
    return 'compiled' if ext() is not None else 'ctypes'


@contextlib.contextmanager
def binding(which: str):
    """Force 'ctypes' or 'compiled' for the prepared launches BUILT inside the block (a launch keeps the binding it was
    prepared with)."""
    if which not in ('ctypes', 'compiled'):
        raise ValueError("'ctypes' or 'compiled'")
    prev = _state['forced']
    _state['forced'] = which
    try:
        if which == 'compiled' and ext() is None:
            raise _ffi.CwnError('the compiled binding is not built: python -m cwn_amd._build_ext')
        yield
    finally:
        _state['forced'] = prev


def fn_address(fn) -> int:
    import ctypes
    return int(ctypes.cast(fn, ctypes.c_void_p).value)
