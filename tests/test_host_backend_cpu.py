"""CPU tests of the host back-end layer (ReprojectionFactor/ReprojectionBatch, ResidualBlockInfo, MarginalizationInfo,
MarginalizationFactor, Preintegration/PreintegrationFactor) linked against the oracle ABI shim."""
import ctypes as C

import backend_utils as bu
from stream_utils import ensure_oracle_host


def _lib():
    return C.CDLL(ensure_oracle_host())


def test_reprojection_costfunction_surface(oracle):
    bu.check_reproj_costfunction_surface(_lib(), oracle)


def test_marginalization_pipeline(oracle):
    bu.check_marginalization(_lib(), oracle)


def test_marginalization_structured_path_equals_dense():
    bu.check_marginalization_paths(_lib())


def test_marginalization_batch_equals_per_window(oracle):
    """the marginalizations of many streams in one pass (MarginalizationBatch) == each window marginalized on its own, bit for bit, and
    every window == the oracle's assembly + Schur complement of that window's parameters"""
    bu.check_marginalization_batch(_lib(), oracle, bitwise=True)


def test_preintegration_factor(oracle):
    bu.check_preintegration(_lib(), oracle)


def test_preintegration_factor_matches_reference_golden():
    bu.check_preintegration_golden(_lib())


def test_marginalization_matches_reference_golden():
    """host MarginalizationInfo/ResidualBlockInfo/MarginalizationFactor (oracle-backed) vs the REFERENCE's own pipeline"""
    import os
    bu.check_marginalization_golden(_lib(), os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "marg_ref_golden.npz"))


# sha256 (first 16 hex digits) of evals + evecs of the PLAIN tred2 / tql2 form of symmetricEigen (rounds 2-4: one column at a time, strided)
# on the matrices of _eigen_cases(): the restructured routine of round 5 (transposed working matrix, interleaved chains, vectorised element-wise
# loops) must reproduce the bit patterns — the marginalization prior of every replay depends on them
_EIGEN_PLAIN_FORM = {(1, "full"): "edd7a54c9f87f586", (2, "full"): "347938436cebf358", (7, "full"): "1ebbeb2c54d16942",
                     (15, "full"): "c35093550c84e838", (16, "half"): "f651dd2bb1433869", (61, "full"): "7e0b058426858d28",
                     (61, "zero3"): "9905dd043c84cd3d", (133, "scaled"): "fd9bb88ba451bdd8", (142, "full"): "fad3b36234dba0f3",
                     (142, "half"): "60c57d9b4b8eaffd", (143, "diag"): "f8fc86d7dea01ef3", (40, "zeros"): "a082dd02fc7ce925"}


def _eigen_cases():
    import numpy as np
    rng = np.random.default_rng(20260925)
    for n, kind in _EIGEN_PLAIN_FORM:
        rank = max(1, n // 2) if kind == "half" else n + 3
        J = rng.integers(-8, 9, size=(rank, n)).astype(np.float64)
        A = J.T @ J  # small integers: exact, whatever BLAS adds them in
        if kind == "zero3":  # zero rows / columns: the scale == 0 branch of the tridiagonalisation
            A[::3, :] = 0
            A[:, ::3] = 0
        if kind == "scaled":  # powers of two: still exact
            sc = 1.0 + (np.arange(n) % 7) * 1024.0
            A = A * sc[:, None] * sc[None, :]
        if kind == "diag":
            A = np.diag(np.diag(A))
        if kind == "zeros":
            A = np.zeros((n, n))
        yield n, kind, np.ascontiguousarray(A)


def test_symmetric_eigen_keeps_the_plain_forms_bit_patterns():
    import hashlib
    import numpy as np
    lib = _lib()
    lib.icgh_symmetric_eigen.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    for n, kind, A in _eigen_cases():
        ev, V = np.zeros(n), np.zeros((n, n))
        assert lib.icgh_symmetric_eigen(n, A.ctypes.data, ev.ctypes.data, V.ctypes.data) == 0
        assert hashlib.sha256(ev.tobytes() + V.tobytes()).hexdigest()[:16] == _EIGEN_PLAIN_FORM[(n, kind)], (n, kind)
        scale = max(1.0, float(np.abs(A).max()))
        assert np.all(np.diff(ev) >= 0)
        assert np.abs(A @ V - V * ev[None, :]).max() <= 1e-11 * scale * n, (n, kind)
        assert np.abs(V.T @ V - np.eye(n)).max() <= 1e-12 * n, (n, kind)
