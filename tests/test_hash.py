"""include/pbdx.h is a C header and its inline block hash (dirty tracking of a host mirror) is what the plug-in runs on the host:
compile a three-line C file against the header with the system compiler and compare pbdx_hash_block with its numpy restatement (the GPU test
test_device_block_hashes_equal_the_host_definition compares the device kernel with the same restatement).  Properties the tracking relies on: a single
changed word changes its block's hash and no other; swapping two different words changes it; an odd 32-bit tail is covered."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from tests import util
from tests.test_plugin import _np_block_hashes

SRC = r'''
#include "pbdx.h"
uint64_t hash_block(const void *base, uint32_t n, uint32_t elem_bytes, uint32_t block) { return pbdx_hash_block(base, n, elem_bytes, block); }
uint32_t num_blocks(uint32_t n) { return pbdx_hash_num_blocks(n); }
'''


@pytest.fixture(scope="module")
def hashlib_c(tmp_path_factory):
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    d = tmp_path_factory.mktemp("hash")
    (d / "h.c").write_text(SRC)
    so = d / "libh.so"
    subprocess.check_call([cc, "-std=c99", "-O2", "-Wall", "-Werror", "-shared", "-fPIC", "-I", os.path.join(util.ROOT, "include"), "-o", str(so), str(d / "h.c")])
    lib = C.CDLL(str(so))
    lib.hash_block.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
    lib.hash_block.restype = C.c_uint64
    lib.num_blocks.argtypes = [C.c_uint32]
    lib.num_blocks.restype = C.c_uint32
    return lib


def _c_hashes(lib, a):
    a = np.ascontiguousarray(a)
    n, eb = len(a), a.dtype.itemsize * (a.shape[1] if a.ndim > 1 else 1)
    return np.array([lib.hash_block(a.ctypes.data, n, eb, b) for b in range(lib.num_blocks(n))], dtype=np.uint64)


@pytest.mark.parametrize("dtype,cols,n", [(np.float32, 3, 5 * 1024 + 137), (np.float64, 3, 2048), (np.float32, 1, 3 * 1024 + 1), (np.float32, 1, 7), (np.float64, 1, 1025)])
def test_c_header_hash_equals_numpy_restatement(hashlib_c, dtype, cols, n):
    rng = np.random.default_rng(n)
    a = rng.standard_normal((n, cols) if cols > 1 else n).astype(dtype)
    h = _c_hashes(hashlib_c, a)
    assert np.array_equal(h, _np_block_hashes(a if cols > 1 else a.reshape(-1, 1)))
    # one changed element: exactly its block's hash changes
    for i in (0, n // 2, n - 1):
        b = a.copy()
        if cols > 1:
            b[i, 1] = np.nextafter(b[i, 1], dtype(10))
        else:
            b[i] = np.nextafter(b[i], dtype(10))
        h2 = _c_hashes(hashlib_c, b)
        diff = np.nonzero(h != h2)[0]
        assert list(diff) == [i // 1024], (i, diff)
    # two elements of one block swapped
    b = a.copy()
    if n > 3:
        b[[1, 2]] = b[[2, 1]]
        assert _c_hashes(hashlib_c, b)[0] != h[0]
