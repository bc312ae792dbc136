"""ctypes wrapper of the plain-C restatement (oracle/vbmc_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_dp = C.POINTER(C.c_double)


def _p(a):
    return a.ctypes.data_as(_dp)


def _host_stamp():
    """The Makefile compiles with -march=native: a library built on another machine (the development container's build travels
    to the GPU box with the snapshot) may use instructions this host lacks, and would be the wrong baseline anyway.  The stamp is
    the CPU model and its flag set."""
    import hashlib

    try:
        txt = open("/proc/cpuinfo").read()
        keep = [ln for ln in txt.split("\n") if ln.startswith(("model name", "flags"))][:2]
        return hashlib.sha256("\n".join(keep).encode()).hexdigest()[:16]
    except OSError:
        return "unknown"


def ensure_built():
    """Build (or rebuild, when the existing build was made on a different CPU) the two libraries for THIS host."""
    build = os.path.join(_HERE, "_build")
    stamp_file = os.path.join(build, "host_stamp")
    stamp = _host_stamp()
    have = open(stamp_file).read().strip() if os.path.exists(stamp_file) else None
    libs = [os.path.join(build, n) for n in ("liboracle.so", "liboracle_omp.so")]
    src = os.path.join(_HERE, "vbmc_oracle.c")
    stale = have != stamp or not all(os.path.exists(p) and os.path.getmtime(p) >= os.path.getmtime(src) for p in libs)
    if stale:
        subprocess.check_call(["make", "-s", "-B", "-C", _HERE])
        with open(stamp_file, "w") as f:
            f.write(stamp)


def load(openmp=False):
    name = "liboracle_omp.so" if openmp else "liboracle.so"
    path = os.path.join(_HERE, "_build", name)
    ensure_built()
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
