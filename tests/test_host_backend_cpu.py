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


def test_preintegration_factor(oracle):
    bu.check_preintegration(_lib(), oracle)
