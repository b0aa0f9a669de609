"""Build PRODUCT-side containers (cwn_amd.complex) from tests/golden/dummy_complexes.npz."""
import torch

from cwn_amd.complex import Cochain, Complex, ComplexBatch
from tests._golden import load, T

KEYS = ('x', 'upper_index', 'lower_index', 'shared_boundaries', 'shared_coboundaries',
        'boundary_index', 'y')


def dummy_complex(name: str, device=None) -> Complex:
    g = load('dummy_complexes.npz')
    dim = int(g[f'{name}/dimension'])
    cochains = []
    for d in range(dim + 1):
        kw = {k: T(g[f'{name}/{d}/{k}']).clone() for k in KEYS if f'{name}/{d}/{k}' in g}
        cochains.append(Cochain(dim=d, **kw))
    y = T(g[f'{name}/y']).clone() if f'{name}/y' in g else None
    cx = Complex(*cochains, y=y)
    return cx.to(device) if device is not None else cx


def dummy_batch(names, max_dim=2, device=None) -> ComplexBatch:
    b = ComplexBatch.from_complex_list([dummy_complex(n) for n in names], max_dim=max_dim)
    return b.to(device) if device is not None else b


def list_names(which: str):
    return [str(n) for n in load('dummy_complexes.npz')[f'lists/{which}']]


def gate(got, ref, what: str = '', tol: float = 1e-5) -> float:
    """The north-star parity gate: max|got - ref| <= tol * max(1, |ref|_inf), with the observed maximum
    printed into the test log (pytest -s / -rP shows it).  `ref` is the oracle (preferably evaluated in
    float64) or a reference-generated fixture."""
    g = got.detach().cpu().double()
    r = ref.detach().cpu().double()
    assert g.shape == r.shape, (what, tuple(g.shape), tuple(r.shape))
    err = float((g - r).abs().max()) if r.numel() else 0.0
    scale = max(1.0, float(r.abs().max())) if r.numel() else 1.0
    # stated precisely: the bound is RELATIVE to |ref|_inf once that exceeds 1; whether the
    # north star's absolute 1e-5 holds as well is printed with every line
    print(f'[gate] {what}: max|delta| = {err:.3e} absolute  |ref|_inf = {scale:.3g}  bound = {tol * scale:.3e} '
          f'({"also within" if err <= tol else "ABOVE"} the absolute {tol:g})')
    assert err <= tol * scale, f'{what}: max|delta| {err:.3e} > {tol * scale:.3e}'
    return err


def deviation(t, ref64) -> float:
    """max |t - ref64| in float64 (0 for empty tensors)."""
    r = ref64.detach().cpu().double()
    return float((t.detach().cpu().double() - r).abs().max()) if r.numel() else 0.0


def to_double(state: dict) -> dict:
    """A state_dict (or any dict of tensors) with its floating tensors in float64: the oracle then runs
    in double precision and the gate measures the product's own rounding, not the checker's."""
    return {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in state.items()}
