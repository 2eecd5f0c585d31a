import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.lib()          # builds oracle/_build/libmirge_oracle.so if it is missing
    return orc


@pytest.fixture(scope="session")
def hostsim():
    """HOST build of the matcher's __host__ __device__ arithmetic (tests/hostsim)."""
    import ctypes as C
    d = os.path.join(ROOT, "tests", "hostsim")
    so = os.path.join(d, "libhostsim.so")
    srcs = [os.path.join(d, "hostsim.hip"), os.path.join(ROOT, "mirge_amd", "csrc", "kernels.h"),
            os.path.join(ROOT, "mirge_amd", "csrc", "keys.h"), os.path.join(ROOT, "mirge_amd", "csrc", "fastq.h"), os.path.join(ROOT, "mirge_amd", "csrc", "trim.h"),
            os.path.join(ROOT, "mirge_amd", "csrc", "ebwt.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["hipcc", "-O2", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, srcs[0]])
    L = C.CDLL(so)
    L.sim_lib_new.restype = C.c_void_p
    L.sim_lib_new.argtypes = [C.c_int32, C.c_char_p, C.c_void_p, C.c_int32, C.c_int32]
    L.sim_lib_free.argtypes = [C.c_void_p]
    L.sim_pack.argtypes = [C.c_char_p, C.c_int32, C.c_void_p]
    L.sim_match.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.sim_match_long.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.sim_fastq.restype = C.c_int64
    L.sim_fastq.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64]
    L.sim_trim_fastq.restype = C.c_int64
    L.sim_trim_fastq.argtypes = [C.c_void_p, C.c_uint64, C.c_char_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    L.sim_count_newlines.restype = C.c_uint64
    L.sim_count_newlines.argtypes = [C.c_void_p, C.c_uint64]
    L.sim_ebwt_extract.restype = C.c_int64
    L.sim_ebwt_extract.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
    L.sim_key_hash.restype = C.c_uint32
    L.sim_key_hash.argtypes = [C.c_void_p, C.c_int32]
    return L
