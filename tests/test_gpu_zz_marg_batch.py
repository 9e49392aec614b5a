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
    """... and EVERY window of every batch against the oracle (orc_reproj + orc_marg on that window's parameters), bit-equal to the per-window
    path: the device assembly forms each sum in an order fixed by the window's factor list (csrc/reproj.hip, k_asm_*)"""
    import oracle_lib
    bu.check_marginalization_batch(C.CDLL(H.HOST_LIB), oracle_lib.load(), bitwise=True)


def test_replay_lockstep_shared_marginalizations_on_gpu(tmp_path):
    """three estimators in lock-step with ICG_LOCKSTEP_MARG_BATCH=1: every marginalization of a tick through one batched launch sequence;
    each stream equals the stream replayed alone bit for bit (identical and different streams, batches of one, two or three windows)"""
    gc.check_replay_lockstep_shared_marginalizations(H.HOST_LIB, tmp_path, bitwise=True)


def test_replay_lockstep_wide_windows_on_gpu(tmp_path):
    """15-keyframe windows (BASELINE configs[3]) through icgh_replay_run_lockstep on the device: every stream equals its own replay alone
    bit for bit"""
    gc.check_replay_lockstep_wide_windows(H.HOST_LIB, tmp_path, bitwise=True)
