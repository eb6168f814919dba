"""tuna_amd -- MI355X-native ERI / Fock-build / SCF engine behind TUNA's integral-module and SCF seams.

    from tuna_amd import integral as ints      # drop-in for tuna_integrals.tuna_integral   (seam 1)
    from tuna_amd import scf                   # drop-in names of tuna_scf                  (seam 2)
    from tuna_amd.energy import run            # "SPE : N N 1.0977 : HF CC-PVTZ"

Every numerical step runs in tuna_amd/libtunafock.so (HIP, gfx950) behind the C ABI of include/tunafock.h; there is no
CPU fallback.
"""
__version__ = "0.1.0"
