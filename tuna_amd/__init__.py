"""tuna_amd -- MI355X-native ERI / Fock-build / SCF engine behind TUNA's integral-module and SCF seams.

    from tuna_amd import integral as ints      # drop-in for tuna_integrals.tuna_integral   (seam 1)
    from tuna_amd import scf                   # drop-in names of tuna_scf                  (seam 2)
    from tuna_amd.energy import run            # "SPE : N N 1.0977 : HF CC-PVTZ"

Every numerical step runs in tuna_amd/libtunafock.so (HIP, gfx950) behind the C ABI of include/tunafock.h; there is no
CPU fallback.
"""
__version__ = "0.1.0"

import os as _os

# Hardware queues of the process: the HIP runtime maps all streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default), and the launches
# of one queue run one after the other, each as long as its slowest workgroup.  A tensor build of a contracted basis set issues ~70
# launches of very different lengths on 8 streams: with 16 queues Ar2/cc-pVQZ's ERI kernels take 8.7 ms instead of 14.7.
# The runtime reads the variable when it initialises, so it has to be in the environment before the first HIP call of the process --
# importing this package before torch touches the GPU is enough; an explicit setting of the caller is left alone.  It is a setting of
# the whole process (torch and RCCL see it too): TUNA_NO_HWQ=1 opts out, and the library itself never touches the environment.
if not _os.environ.get("TUNA_NO_HWQ"):
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


def cpu_quota() -> int:
    """CPUs this process may actually use: the cgroup quota if there is one, else the affinity mask."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            return max(1, int(int(quota) / int(period)))
    except (OSError, ValueError):
        pass
    try:
        return max(1, len(_os.sched_getaffinity(0)))
    except AttributeError:
        return _os.cpu_count() or 1


def limit_host_threads(n: int | None = None):
    """Caps the thread pool of NumPy's BLAS.  The host side of this package only multiplies matrices of a few hundred rows
    (guess projection, Mulliken populations, the host-orchestrated SCF loops), but OpenBLAS starts one thread per visible core --
    256 on an MI355X host -- and its idle workers spin after every call.  In a container with a CPU quota (16 CPUs on the GPU
    boxes) that spinning exhausts the quota and the kernel throttles the whole process, HIP runtime threads included: measured
    50-70 ms stalls inside hipDeviceSynchronize / hipMemcpy, an Ar2/cc-pVQZ single point 100-160 ms instead of 54 ms.
    Default: min(4, CPU quota); TUNA_AMD_HOST_BLAS_THREADS=<n> overrides, 0 leaves the pool alone."""
    if n is None:
        env = _os.environ.get("TUNA_AMD_HOST_BLAS_THREADS")
        n = int(env) if env is not None else min(4, cpu_quota())
    if n <= 0:
        return None
    # BLAS libraries loaded later (SciPy ships its own OpenBLAS) read their pool size from the environment when they start
    _os.environ.setdefault("OPENBLAS_NUM_THREADS", str(n))
    try:
        import numpy  # noqa: F401  (the BLAS library has to be loaded before threadpoolctl can find it)
        from threadpoolctl import threadpool_limits
        return threadpool_limits(limits=n)
    except Exception:
        return None


_host_threads = limit_host_threads()
