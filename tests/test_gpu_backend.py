"""GPU tests of the back-end through the reference's API surface (ceres::CostFunction::Evaluate per factor,
MarginalizationInfo::marginalization, PreintegrationFactor::Evaluate) with the HIP library underneath."""
import ctypes as C

import pytest

import backend_utils as bu
import harness as H

pytestmark = pytest.mark.gpu


def _lib():
    return C.CDLL(H.HOST_LIB)


def test_reprojection_costfunction_surface(oracle):
    bu.check_reproj_costfunction_surface(_lib(), oracle)


def test_marginalization_pipeline(oracle):
    bu.check_marginalization(_lib(), oracle)


def test_marginalization_structured_path_equals_dense():
    bu.check_marginalization_paths(_lib())


def test_preintegration_factor(oracle):
    bu.check_preintegration(_lib(), oracle)


def test_preintegration_factor_matches_reference_golden():
    bu.check_preintegration_golden(_lib())


def test_marginalization_matches_reference_golden():
    """host MarginalizationInfo with the GPU-evaluated / GPU-assembled reprojection factors vs the REFERENCE's own pipeline"""
    import os
    bu.check_marginalization_golden(_lib(), os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "marg_ref_golden.npz"))
