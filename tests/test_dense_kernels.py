"""host/dense_kernels.cc (the reduced-system Cholesky solve and the J^T J accumulation of the window solvers) against numpy, through the C
entries of the host library on the CPU shim (same source as the product's libicgvins_host.so)."""
import ctypes as C

import numpy as np
import pytest

from stream_utils import ensure_oracle_host


def _lib():
    lib = C.CDLL(ensure_oracle_host())
    lib.icgh_dense_cholesky_solve.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    lib.icgh_dense_accumulate_jtj.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.icgh_dense_accumulate_jtj.restype = None
    return lib


@pytest.mark.parametrize("n", [1, 2, 7, 8, 9, 16, 17, 67, 157])
def test_cholesky_solve_matches_numpy(n):
    lib = _lib()
    rng = np.random.RandomState(n)
    M = rng.randn(n + 5, n)
    A = M.T @ M + 1e-3 * np.eye(n)
    b = rng.randn(n)
    A_in = np.ascontiguousarray(np.tril(A) + np.triu(np.full((n, n), np.nan), 1))  # the upper triangle must not be read
    x = b.copy()
    assert lib.icgh_dense_cholesky_solve(n, A_in.ctypes.data, x.ctypes.data) == 0
    ref = np.linalg.solve(A, b)
    assert np.max(np.abs(x - ref)) <= 1e-9 * max(1.0, np.max(np.abs(ref)))
    L = np.linalg.cholesky(A)
    assert np.max(np.abs(np.tril(A_in) - L)) <= 1e-10 * np.max(np.abs(L))  # the factor is left in the lower triangle


def test_cholesky_rejects_indefinite_and_nan():
    lib = _lib()
    A = np.ascontiguousarray(np.array([[1.0, 0.0], [2.0, 1.0]]))  # symmetric completion [[1,2],[2,1]] is indefinite
    b = np.ones(2)
    assert lib.icgh_dense_cholesky_solve(2, A.ctypes.data, b.ctypes.data) == -1
    A = np.ascontiguousarray(np.array([[np.nan]]))
    b = np.ones(1)
    assert lib.icgh_dense_cholesky_solve(1, A.ctypes.data, b.ctypes.data) == -1


@pytest.mark.parametrize("nr,nf", [(1, 1), (2, 6), (15, 32), (6, 7), (142, 142), (3, 9)])
def test_accumulate_jtj_is_the_cell_by_cell_inner_product(nr, nf):
    """every cell equals sum_k J[k,x] J[k,y] accumulated in ascending k from zero — bit for bit (the order the per-cell loop of the reference's
    ResidualBlockInfo accumulation has), upper triangle only, on top of what T held"""
    lib = _lib()
    rng = np.random.RandomState(nr * 100 + nf)
    J = np.ascontiguousarray(rng.randn(nr, nf))
    r = rng.randn(nr)
    T0 = rng.randn(nf, nf)
    g0 = rng.randn(nf)
    T, g = T0.copy(), g0.copy()
    lib.icgh_dense_accumulate_jtj(nr, nf, J.ctypes.data, r.ctypes.data, T.ctypes.data, g.ctypes.data)
    ref_T, ref_g = T0.copy(), g0.copy()
    for k in range(nr):  # T += row-by-row rank-one updates: per cell ((T + p0) + p1) + ... in ascending k
        ref_T += np.triu(np.outer(J[k], J[k]))
        ref_g += J[k] * r[k]
    assert np.array_equal(np.triu(T), np.triu(ref_T))
    assert np.array_equal(np.tril(T, -1), np.tril(T0, -1))  # the lower triangle is not touched
    assert np.array_equal(g, ref_g)


def test_block_pool_cross_thread_traffic():
    """host/object_pool.h: blocks allocated on one thread and freed on another (spill / refill of the per-thread lists), two size classes,
    LIFO reuse — no block is handed out twice or loses its contents"""
    lib = C.CDLL(ensure_oracle_host())
    lib.icgh_pool_selftest.argtypes = [C.c_int, C.c_int]
    for threads, iters in [(1, 5000), (4, 40000), (7, 30001)]:
        assert lib.icgh_pool_selftest(threads, iters) == 0, (threads, iters)
