import sys, ctypes, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aleo_amd
from aleo_amd import synth
B = synth.g2_multiples_affine200(64)
rows = np.ascontiguousarray(B[:, :192])
f = (ctypes.c_uint32*2)()
rc = aleo_amd.lib().aleo_mi355x_selftest_g2pair(rows.ctypes.data_as(ctypes.c_void_p), 64, 62, f)
print('rc', rc, 'failures', f[0], 'steps', f[1])
