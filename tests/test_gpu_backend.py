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


def test_preintegration_factor(oracle):
    bu.check_preintegration(_lib(), oracle)
