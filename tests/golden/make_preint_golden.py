#!/usr/bin/env python3
"""Generates tests/golden/preint_ref_golden.npz by running the REFERENCE's IMU preintegration
(oracle/_ref/libref_preint.so = /root/reference/.../preintegration/preintegration_{base,normal,earth}.{h,cc} and
preintegration_factor.h compiled unmodified against the interface shims in oracle/ref_build/shim).  Run in the build
container only (needs /root/reference):
    make -C oracle/ref_build && python tests/golden/make_preint_golden.py
Per case: IMU interval, start state, parameters (+ station for the Earth variant) -> current/delta state, 15x15 Jacobian and
covariance, delta time, Earth rate, and PreintegrationFactor::Evaluate (residual 15, Jacobians 15x7, 15x9, 15x7, 15x9) at a
perturbed end state."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import preint_data as pd  # noqa: E402

STATION = np.array([30.5 * np.pi / 180, 114.3 * np.pi / 180, 20.0])


def run_case(lib, variant, imu, s0, s1_offset, station):
    p = lambda a: np.ascontiguousarray(a, np.float64).ctypes.data_as(C.c_void_p)
    cur, delta, jac, cov = np.zeros(16), np.zeros(16), np.zeros((15, 15)), np.zeros((15, 15))
    dt, iewn = C.c_double(), np.zeros(3)
    if variant == 0:
        assert lib.ref_preint_integrate(len(imu), p(imu), p(s0), p(pd.PARAMS), p(cur), p(delta), p(jac), p(cov), C.byref(dt)) == 0
    else:
        assert lib.ref_preint_integrate_earth(len(imu), p(imu), p(s0), p(pd.PARAMS), p(station), p(cur), p(delta), p(jac), p(cov),
                                              C.byref(dt), p(iewn)) == 0
    s1 = cur + s1_offset
    pose0, mix0 = pd.split(s0)
    pose1, mix1 = pd.split(s1)
    r, J = np.zeros(15), np.zeros(480)
    if variant == 0:
        assert lib.ref_preint_factor(len(imu), p(imu), p(s0), p(pd.PARAMS), p(pose0), p(mix0), p(pose1), p(mix1), p(r), p(J)) == 0
    else:
        assert lib.ref_preint_factor_earth(len(imu), p(imu), p(s0), p(pd.PARAMS), p(station), p(pose0), p(mix0), p(pose1), p(mix1),
                                           p(r), p(J)) == 0
    return dict(cur=cur, delta=delta, jac=jac, cov=cov, dt=dt.value, iewn=iewn, s1=s1, r=r, J=J)


def main():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_preint.so"))
    out = {"station": STATION, "params": pd.PARAMS}
    k = 0
    for variant in (0, 1):
        for seed, n in ((0, 41), (1, 21), (2, 81), (3, 6)):
            imu = pd.make_interval(n=n, seed=100 * variant + seed)
            s0 = pd.state(rv=(0.02 + 0.1 * seed, -0.03, 0.4 - 0.2 * seed))
            off = np.zeros(16)
            off[:3] = [0.01, -0.02, 0.005]
            off[3:7] = [1e-3, -2e-3, 5e-4, 0.0]  # non-unit end quaternion (Ceres hands over whatever Plus() produced)
            off[7:10] = 0.01
            off[10:13] = 1e-5
            off[13:] = 1e-4
            # SURVEY.md hazard H9: the reference never assigns IntegrationParameters::station, so in the shipped system it is
            # (0, 0, 0) (value-initialised); half of the Earth cases use that effective value, half a real origin
            station = np.zeros(3) if seed >= 2 else STATION
            c = run_case(lib, variant, imu, s0, off, station)
            out.update({f"c{k}_variant": variant, f"c{k}_imu": imu, f"c{k}_s0": s0})
            out.update({f"c{k}_{name}": val for name, val in c.items()})
            k += 1
    out["n_cases"] = k
    np.savez(os.path.join(ROOT, "tests", "golden", "preint_ref_golden.npz"), **out)
    print("wrote", k, "cases")


if __name__ == "__main__":
    main()
