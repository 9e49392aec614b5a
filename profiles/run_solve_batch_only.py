import sys, ctypes as C
sys.path.insert(0,"tests"); sys.path.insert(0,"ic-gvins_amd")
import solve_utils as su, harness as H
hl = C.CDLL(H.HOST_LIB)
Pz = su.make_problem(300, 10, seed=4, n_outliers=10, perturb=0.2)
su.host_solve_batch(hl, [Pz]*4)
for _ in range(3):
    res, ms = su.host_solve_batch(hl, [Pz]*256)
print(ms)
