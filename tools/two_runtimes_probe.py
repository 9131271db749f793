"""Two HIP runtimes in one process (PyTorch-ROCm bundles its own libamdhip64.so; libmsfm_match.so links the system one): which order of
initialisation works?  python tools/two_runtimes_probe.py [libmsfm_match.so]   -> profiles/r05_two_hip_runtimes.txt"""
import subprocess
import sys

LIB = sys.argv[1] if len(sys.argv) > 1 else "monocularsfm_amd/csrc/libmsfm_match.so"
CREATE = "import ctypes as C; L = C.CDLL(%r); h = C.c_void_p(); L.msfm_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]; rc = L.msfm_create(0, C.byref(h))" % LIB
TORCH = """
import torch
try:
    torch.zeros(4, device="cuda"); t = "ok"
except Exception as e:
    t = "FAILED: " + str(e)[:60]
"""
for name, code in (("msfm_create first, then torch", CREATE + TORCH + "\nprint('msfm_create rc', rc, '| torch', t)"),
                   ("torch first, then msfm_create", TORCH + "\n" + CREATE + "\nprint('torch', t, '| msfm_create rc', rc)")):
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    print("%-32s: %s" % (name, (r.stdout.strip().splitlines() or [r.stderr.strip()[-200:]])[-1]), flush=True)
