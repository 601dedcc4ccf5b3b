"""aleo_amd — MI355X-native backend for the Aleo SDK's execute/prove hot path (BLS12-377 G1 MSM + Fr NTT).

Product code: HIP kernels + C ABI in aleo_amd/csrc (built to aleo_amd/lib/libaleo_mi355x.so) and the host-side
mirrors of the reference's operator interfaces (msm.VariableBase, fft.EvaluationDomain, kzg.KZG10)."""
import os as _os
# The HIP runtime maps a process's streams onto at most GPU_MAX_HW_QUEUES hardware queues (4 by default) and reads the variable when it initialises.  The
# library runs a caller's stream, a side stream, high-priority streams and the prover's worker streams at once: with 8 queues the small launches of
# lockstep proofs stop sharing three queues (2^15 constraints, 8 proofs per call: 35.6 -> 33.7 ms; 16 is slower).  A value the user has set wins.
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
from ._lib import lib, LIB_PATH, EXPORTS, AleoMi355xError, UnsatisfiedAssignment          # noqa: F401
from .msm import VariableBase, PinnedBases, ShardedBases, g1_sum, last_msm_timing  # noqa: F401
from .fft import EvaluationDomain                                   # noqa: F401
from .kzg import KZG10, SonicKZG10, CommitterKey                    # noqa: F401
