"""GPU test of MarginalizationBatch (host/marg_batch.h): the marginalizations of many streams in one pass == each window marginalized
on its own by MarginalizationInfo::marginalization() (reference factors/marginalization_info.h:73-101), on the HIP library.
(The file sorts after the other GPU tests on purpose: the class is the newest code of the round.)"""
import ctypes as C

import pytest

import backend_utils as bu
import gvins_checks as gc
import harness as H

pytestmark = pytest.mark.gpu


def test_marginalization_batch_equals_per_window():
    bu.check_marginalization_batch(C.CDLL(H.HOST_LIB))


def test_replay_lockstep_shared_marginalizations_on_gpu(tmp_path):
    """three estimators in lock-step with ICG_LOCKSTEP_MARG_BATCH=1: every marginalization of a tick through one batched launch sequence;
    each stream equals the stream replayed alone to the rounding of the FP64-atomic assembly"""
    gc.check_replay_lockstep_shared_marginalizations(H.HOST_LIB, tmp_path, bitwise=False)


def test_replay_lockstep_wide_windows_on_gpu(tmp_path):
    """15-keyframe windows (BASELINE configs[3]) through icgh_replay_run_lockstep on the device: every stream equals its own replay alone to
    the rounding of the FP64-atomic assembly, whichever path (batched or the estimator's own solver) takes the wide windows"""
    gc.check_replay_lockstep_wide_windows(H.HOST_LIB, tmp_path, bitwise=False)
