"""ctypes wrapper of the plain-C restatement (oracle/vbmc_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_dp = C.POINTER(C.c_double)


def _p(a):
    return a.ctypes.data_as(_dp)


def load(openmp=False):
    name = "liboracle_omp.so" if openmp else "liboracle.so"
    path = os.path.join(_HERE, "_build", name)
    if not os.path.exists(path):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    lib = C.CDLL(path)
    lib.oracle_num_threads.restype = C.c_int
    if openmp:
        lib.oracle_set_threads(C.c_int(usable_cores()))
    return lib


def usable_cores():
    """Cores this process may really use: the cgroup CPU quota if there is one (a container that sees 256 logical CPUs but is
    granted 16 only thrashes with a 256-thread team), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, -(-q // per)))
        except (OSError, ValueError):
            pass
    return n


def negelcbo(theta, X, hyp, alpha, eps, meanfun=4, Nnoise=1, grad=True, openmp=False):
    """(F, dF, G, H) for all four groups optimised, beta = 0, no soft bounds.
    eps: (K, Mh, D) C-order == D x Mh x K column-major."""
    lib = load(openmp)
    X = np.asfortranarray(X, dtype=np.float64)
    N, D = X.shape
    hyp = np.asfortranarray(hyp, dtype=np.float64)
    Nhyp, S = hyp.shape
    alpha = np.asfortranarray(alpha, dtype=np.float64)
    eps = np.ascontiguousarray(eps, dtype=np.float64)
    K, Mh, _ = eps.shape
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    T = D * K + K + D + K
    assert theta.size == T
    dF = np.zeros(T)
    F, G, H = C.c_double(), C.c_double(), C.c_double()
    lib.oracle_negelcbo(C.c_int(D), C.c_int(K), C.c_int(N), C.c_int(S), C.c_int(Nhyp), C.c_int(Nnoise), C.c_int(meanfun),
                        C.c_int(Mh), _p(theta), _p(X), _p(hyp), _p(alpha), _p(eps), C.c_int(1 if grad else 0), C.byref(F), _p(dF),
                        C.byref(G), C.byref(H))
    return F.value, dF, G.value, H.value
