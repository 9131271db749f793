"""Two HIP runtimes in one process (PyTorch-ROCm bundles its own libamdhip64.so; libmsfm_match.so links the system one): which order of
initialisation works?  python tools/two_runtimes_probe.py <libmsfm_match.so>   -> profiles/r05_two_hip_runtimes.txt"""
import ctypes as C, sys
path = sys.argv[1]
L = C.CDLL(path)
h = C.c_void_p()
L.msfm_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
print("create rc", L.msfm_create(0, C.byref(h)))
import torch
try:
    t = torch.zeros(4, device="cuda"); print(path, "torch ok")
except Exception as e:
    print(path, "torch FAILED", str(e)[:60])
