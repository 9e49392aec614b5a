"""Synthetic IMU intervals (200 Hz) for the preintegration tests: analytic rates + noise per config/gvins.yaml:26-31."""
import numpy as np

D2R = np.pi / 180.0
# imumodel of config/gvins.yaml converted as GVINS does (ic_gvins.cc:94-99): arw deg/sqrt(hr), vrw m/s/sqrt(hr),
# gbstd deg/hr, abstd mGal, corrtime hr
PARAMS = np.array([0.1 * D2R / 60.0, 0.1 / 60.0, 50.0 * D2R / 3600.0, 50.0 * 1e-5, 1.0 * 3600.0, 9.8, 7.292115e-5, 0.0, 0.0])


def make_interval(n=41, rate=200.0, seed=0, noise=True, omega=(0.02, -0.05, 0.2), acc=(0.3, -0.2, -9.7)):
    rng = np.random.RandomState(seed)
    dt = 1.0 / rate
    imu = np.zeros((n, 8))
    for k in range(n):
        imu[k, 0] = 1000.0 + k * dt
        imu[k, 1] = dt
        w = np.array(omega) * (1 + 0.3 * np.sin(0.7 * k * dt))
        a = np.array(acc) + np.array([0.5 * np.sin(3 * k * dt), 0.2 * np.cos(2 * k * dt), 0.0])
        imu[k, 2:5] = w * dt + (rng.normal(0, 1e-5, 3) if noise else 0)
        imu[k, 5:8] = a * dt + (rng.normal(0, 1e-4, 3) if noise else 0)
    return imu


def state(p=(1.0, 2.0, -0.5), rv=(0.02, -0.03, 0.4), v=(3.0, 0.2, -0.1), bg=(1e-4, -2e-4, 5e-5), ba=(1e-3, 2e-3, -1e-3)):
    rv = np.array(rv)
    a = np.linalg.norm(rv)
    q = np.array([*(np.sin(a / 2) * rv / a), np.cos(a / 2)]) if a > 0 else np.array([0, 0, 0, 1.0])
    return np.concatenate([p, q, v, bg, ba])


def split(state16):
    """-> pose[7] (p, qxyzw), mix[9] (v, bg, ba)"""
    return state16[:7].copy(), state16[7:].copy()


def bench_block(icgvins, device, n_streams=256, n_intervals=15, n_samples=40, cpu=None):
    """bench.py's C4 preintegration leg: n_streams x n_intervals intervals of n_samples 200 Hz samples in ONE icg_preint_batch launch
    (Earth variant, the more expensive one); `cpu` = an oracle_lib.Oracle to time the same interval on one host core."""
    import time
    base = [make_interval(n_samples + 1, seed=s) for s in range(n_intervals)]
    imu = np.concatenate(base * n_streams)
    off = (np.arange(n_streams * n_intervals + 1) * (n_samples + 1)).astype(np.int32)
    s0 = np.tile(state(), (n_streams * n_intervals, 1))
    ctx = icgvins.Context(640, 480, n_slots=1, max_batch=1, max_points=64, device=device)
    for _ in range(2):
        ctx.preint_batch(1, off, imu, s0, PARAMS)
    ctx.prof_enable(True)
    nrep = 5
    t1 = time.perf_counter()
    for _ in range(nrep):
        ctx.preint_batch(1, off, imu, s0, PARAMS)
    wall = (time.perf_counter() - t1) / nrep
    n_launch, ms = ctx.prof()["preint"]
    ks = ms * 1e-3 / n_launch
    nsamp = n_streams * n_intervals * n_samples
    out = {"metric": "preintegration samples/s (P1: PreintegrationEarth::integrationProcess + Jacobian/covariance propagation)",
           "intervals_per_launch": n_streams * n_intervals, "samples_per_interval": n_samples, "value": round(nsamp / ks, 1), "unit": "samples/s",
           "kernel_us": round(ks * 1e6, 1), "call_wall_us_incl_transfers": round(wall * 1e6, 1),
           "bound": "latency: one wave per interval, strictly sequential over its samples (72 B in per sample, 15x15 J and P in LDS)"}
    ctx.close()
    if cpu is not None:
        t1 = time.perf_counter()
        nloop = 0
        while time.perf_counter() - t1 < 2.0:
            cpu.preint_integrate(1, base[nloop % n_intervals], state(), PARAMS)
            nloop += 1
        out["cpu_baseline"] = {"value": round(nloop * n_samples / (time.perf_counter() - t1), 1), "unit": "samples/s", "cores": 1, "kind": "port",
                               "sample": f"{nloop} x one {n_samples}-sample interval, oracle (-O2), 1 thread"}
    return out
