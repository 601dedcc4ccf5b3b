"""tools/prof_workload.py against another build of the library (an A/B of a compile-time variant on the same box, not a test):
python tools/ab_lib.py <path to a libaleo_mi355x.so | -> <workload> [reps]   ('-' = the in-tree build)."""
import sys, os, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import aleo_amd._lib as L
if sys.argv[1] != '-': L.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, 'tools', 'prof_workload.py')] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name='__main__')
