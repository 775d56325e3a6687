"""The C-ABI library must build, load and export every symbol include/smaltgpu.h declares
(no compute calls here: this runs without a GPU)."""
import ctypes
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_declared_symbols():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "smalt_amd", "csrc")], check=True)
    hdr = open(os.path.join(ROOT, "include", "smaltgpu.h")).read()
    declared = sorted(set(re.findall(r"\b(smaltgpu_[a-z_]+)\s*\(", hdr)))
    assert len(declared) >= 15
    lib = ctypes.CDLL(os.path.join(ROOT, "smalt_amd", "libsmaltgpu.so"))
    for name in declared:
        assert hasattr(lib, name), name


def test_no_device_fails_loudly():
    from smalt_amd import api
    if api.device_count() > 0:
        return
    try:
        api.Index.load("/nonexistent/prefix", 0)
    except api.SmaltGpuError:
        return
    raise AssertionError("loading an index without files/devices must raise")
