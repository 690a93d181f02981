"""The DCT of the path against real FFTW output (CPU; SURVEY.md §8(c), VERDICT r1 "next" #2).

The reference's only un-vendored arithmetic is FFTW's single-precision REDFT10 / REDFT01
(src/nlkalman.c:204-220, 278, 355), absent from this image. tests/golden/fftw_single_dct.npz
holds FFTW-computed single-precision transforms of x = 0..n-1 (provenance:
tests/golden/make_fftw_pin.py). Here the oracle's basis (oracle/nlk_oracle.c:nlko_dct_basis), the
basis the product uploads to the device (csrc/nlk_hip.hip:host_basis) and the matrix that the 12-point
flow graph of the packed-lane kernel applies (csrc/k_dct12.h) are each driven with that input, the
reference's scaling (src/nlkalman.c:281-298 forward, :335-353 inverse) is undone / replayed in
float, and the result must agree with FFTW's to <= 2 ulp of the vector's largest element.

This pins FFTW's transform (what the reference calls), not the reference binary."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fftw_single_dct.npz")
SIZES = (4, 8, 12, 16)
F = np.float32


def _seq_dot(a, b):
    """float32 dot product accumulated left to right, one rounding per op (the oracle's loop)."""
    acc = F(0)
    for x, y in zip(a, b):
        acc = F(acc + F(F(x) * F(y)))
    return acc


def _scales(n):
    return np.array([np.sqrt(1.0 / n)] + [np.sqrt(2.0 / n)] * (n - 1))


def _check_basis(C, n, gold, what):
    C = np.asarray(C, np.float32)
    x = np.arange(n, dtype=np.float32)
    s = _scales(n)
    # REDFT10: Y_k = 2 sum_j x_j cos(pi (j + 1/2) k / n) = (2 / s_k) (C x)_k
    y = np.array([_seq_dot(x, C[k]) for k in range(n)], np.float32)
    ref = gold[f"redft10_{n}"]
    ulp = np.spacing(F(np.abs(ref).max()))
    err = np.abs(2.0 / s * y.astype(np.float64) - ref.astype(np.float64)).max()
    assert err <= 2 * ulp, f"{what}: forward n={n}: {err / ulp:.2f} ulp"
    # REDFT01: Y_k = x_0 + 2 sum_{j>=1} x_j cos(pi j (k + 1/2) / n) = sum_j C[j][k] x_j (1/s_0 | 2/s_j)
    xin = (x.astype(np.float64) * np.where(np.arange(n) == 0, 1.0 / s, 2.0 / s)).astype(np.float32)
    z = np.array([_seq_dot(xin, C[:, k]) for k in range(n)], np.float32)
    ref = gold[f"redft01_{n}"]
    ulp = np.spacing(F(np.abs(ref).max()))
    err = np.abs(z.astype(np.float64) - ref.astype(np.float64)).max()
    assert err <= 2 * ulp, f"{what}: inverse n={n}: {err / ulp:.2f} ulp"


@pytest.fixture(scope="module")
def gold():
    with np.load(GOLD) as g:
        return {k: g[k] for k in g.files}


def test_fixture_is_fftws_definition(gold):
    """Sanity of the fixture itself: FFTW's published definitions, in double."""
    for n in SIZES:
        x, j = np.arange(n, dtype=np.float64), np.arange(n)
        r10 = np.array([2 * np.sum(x * np.cos(np.pi * (j + 0.5) * k / n)) for k in range(n)])
        r01 = np.array([x[0] + 2 * np.sum(x[1:] * np.cos(np.pi * j[1:] * (k + 0.5) / n)) for k in range(n)])
        assert np.abs(gold[f"redft10_{n}"] - r10).max() <= 2 * np.spacing(F(np.abs(r10).max()))
        assert np.abs(gold[f"redft01_{n}"] - r01).max() <= 2 * np.spacing(F(np.abs(r01).max()))


def test_oracle_basis_against_fftw(O, gold):
    for n in SIZES:
        _check_basis(O.dct_basis(n), n, gold, "oracle basis")


def test_device_tables_against_fftw_and_oracle(built, O, gold):
    """What upload_tables() sends to the GPU, bit for bit the oracle's basis and window, and
    FFTW-pinned the same way; likewise the matrix that the 12-point flow graph of the packed-lane kernel
    applies (csrc/k_dct12.h, compiled for the host by nlk_host_tables)."""
    for n in SIZES:
        b, w, b12 = built.host_tables(n)
        assert np.array_equal(b, O.dct_basis(n)), f"device basis {n} != oracle basis"
        assert np.array_equal(w, O.window(n)), f"device window {n} != oracle window"
        _check_basis(b, n, gold, "device basis")
    # (a flow graph rounds differently from the table: entries within 2 ulp of the largest one)
    assert np.abs(b12 - O.dct_basis(12)).max() <= 2 * np.spacing(F(0.41))
    _check_basis(b12, 12, gold, "k_dct12.h flow graph")


def test_oracle_2d_transform_replays_reference_scaling_of_fftw(O, gold):
    """The oracle's compiled 2-D transform on the separable input X = x (x) x against what the
    reference computes from FFTW's output: the 3-D REDFT10 plan of size {1, n, n}
    (src/nlkalman.c:204-212) returns 2 * D_u * D_v (D = 1-D REDFT10 of x; the factor 2 is the
    length-1 dimension), then `norm = 1/sqrt(8 n n)` and `1/sqrt(2)` on row 0, column 0 and the
    whole single plane (:281-298) in float. Inverse: the same scaling the other way (:335-353)
    then REDFT01, checked as a round trip of the forward result."""
    for n in SIZES:
        x = np.arange(n, dtype=np.float32)
        X = np.outer(x, x).astype(np.float32)
        D = gold[f"redft10_{n}"]
        fft = (F(2) * np.outer(D, D)).astype(np.float32)           # FFTW's output, to its rounding
        norm = F(1.0 / np.sqrt(8.0 * float(F(n * n))))
        isqrt2 = F(1.0 / np.sqrt(2.0))
        want = (fft * norm).astype(np.float32)
        want[:, 0] = want[:, 0] * isqrt2                             # x == 0 column
        want[0, :] = want[0, :] * isqrt2                             # y == 0 row
        want = (want * isqrt2).astype(np.float32)                    # t == 0 plane: all of it
        got = O.dct2(X[None])[0]
        ulp = np.spacing(F(np.abs(want).max()))
        assert np.abs(got - want).max() <= 4 * ulp, f"2-D forward n={n}: {np.abs(got - want).max() / ulp:.2f} ulp"
        back = O.dct2(want[None].copy(), inverse=True)[0]
        ulpx = np.spacing(F(np.abs(X).max()))
        assert np.abs(back - X).max() <= 4 * ulpx, f"2-D inverse n={n}: {np.abs(back - X).max() / ulpx:.2f} ulp"
