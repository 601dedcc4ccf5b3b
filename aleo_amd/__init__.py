"""aleo_amd — MI355X-native backend for the Aleo SDK's execute/prove hot path (BLS12-377 G1 MSM + Fr NTT).

Product code: HIP kernels + C ABI in aleo_amd/csrc (built to aleo_amd/lib/libaleo_mi355x.so) and the host-side
mirrors of the reference's operator interfaces (msm.VariableBase, fft.EvaluationDomain, kzg.KZG10)."""
from ._lib import lib, LIB_PATH, EXPORTS, AleoMi355xError, UnsatisfiedAssignment          # noqa: F401
from .msm import VariableBase, PinnedBases, ShardedBases, g1_sum, last_msm_timing  # noqa: F401
from .fft import EvaluationDomain                                   # noqa: F401
from .kzg import KZG10, SonicKZG10, CommitterKey                    # noqa: F401
