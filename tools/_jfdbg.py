import os, sys, ctypes
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
sys.argv = ["x"]
import runpy
runpy.run_path(os.path.join(os.environ["GRAFT_REPO_ROOT"], "tools/bench_joint.py"), run_name="__main__")
from indic_cl_asr_amd import _lib
L = _lib.lib()
buf = (ctypes.c_ulonglong * 16)()
L.ia_jf_debug_read.argtypes = [ctypes.c_void_p]
L.ia_jf_debug_read(buf)
for w in range(4): print("wave", w, "main loop", buf[2*w], "epilogue", buf[2*w+1])
print("wave 0: issue-W-loads", buf[8], "compute(2 ks)", buf[9], "barrier1", buf[10], "store+barrier2", buf[11])
