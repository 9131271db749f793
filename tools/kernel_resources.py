"""Register / LDS / scratch use of every kernel in the built library, read from the gfx950 code object's own metadata
(llvm-readelf --notes): the numbers the ISA was compiled to.  rocprofv3's `vgpr_count` field is a different quantity (the
dispatch packet's allocation-granule field), which is why tools/rocprof_summary.py prints both.
Usage: python tools/kernel_resources.py [library.so]   ->  name-sorted table; importable: resources(path) -> dict."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = os.path.join(ROOT, "monocularsfm_amd", "csrc", "libmsfm_match.so")
FIELDS = (".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count",
          ".group_segment_fixed_size", ".private_segment_fixed_size", ".max_flat_workgroup_size")


def resources(path=DEFAULT):
    tmp = tempfile.mkdtemp(prefix="msfm_co_")
    try:
        lib = os.path.join(tmp, "lib.so")
        shutil.copy(path, lib)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", lib], check=True, capture_output=True, cwd=tmp)
        co = [f for f in os.listdir(tmp) if "gfx950" in f]
        if not co:
            raise RuntimeError("no gfx950 code object in " + path)
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, co[0])],
                               check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out = {}
    # one "- .agpr_count: ..." block per kernel inside amdhsa.kernels
    for block in re.split(r"\n\s+- \.", notes.split("amdhsa.kernels:")[1].split("amdhsa.target:")[0]):
        block = "." + block.lstrip(" -.")
        kv = dict(re.findall(r"(\.[a-z_]+):\s+'?([^'\n]+)'?", block))
        if ".name" not in kv:
            continue
        out[kv[".name"]] = {f[1:]: int(kv[f]) for f in FIELDS if f in kv and kv[f].isdigit()}
    return out


def demangle(names):
    exe = shutil.which("c++filt") or shutil.which("llvm-cxxfilt", path=LLVM)
    if not exe:
        return list(names)
    r = subprocess.run([exe], input="\n".join(names), capture_output=True, text=True)
    return r.stdout.split("\n")[:len(names)] if r.returncode == 0 else list(names)


def main():
    res = resources(sys.argv[1] if len(sys.argv) > 1 else DEFAULT)
    names = sorted(res)
    print("# vgpr agpr sgpr vgpr_spill sgpr_spill lds_bytes scratch_bytes  kernel")
    for n, d in zip(names, demangle(names)):
        r = res[n]
        print("%5d %4d %4d %6d %6d %8d %7d  %s" % (
            r.get("vgpr_count", -1), r.get("agpr_count", 0), r.get("sgpr_count", -1), r.get("vgpr_spill_count", 0),
            r.get("sgpr_spill_count", 0), r.get("group_segment_fixed_size", 0), r.get("private_segment_fixed_size", 0),
            d.split("(")[0]))


if __name__ == "__main__":
    main()
