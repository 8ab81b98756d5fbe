"""The C-ABI library loads without a GPU and exports every symbol include/nicer_slam_amd.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "nicer_slam_amd.h")).read()
    declared = set(re.findall(r"\b(nsa_[a-z0-9_]+)\s*\(", hdr))
    assert {"nsa_hash_encode_forward", "nsa_hash_encode_backward", "nsa_hash_encode_second_backward"} <= declared
    lib = ctypes.CDLL(os.path.join(ROOT, "nicer_slam_amd", "lib", "libnicer_slam_amd.so"))
    for name in sorted(declared):
        assert hasattr(lib, name), name
    lib.nsa_strerror.restype = ctypes.c_char_p
    assert lib.nsa_strerror(1) == b"GridEncoding: C must be 1, 2, 4, or 8."
    assert lib.nsa_version() >= 1


def test_native_binding_lists_all_exports():
    from nicer_slam_amd import _native
    hdr = open(os.path.join(ROOT, "include", "nicer_slam_amd.h")).read()
    declared = set(re.findall(r"\b(nsa_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_native.EXPORTS)
