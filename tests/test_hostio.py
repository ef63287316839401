"""The library's host-side copy (csrc/pbdx_hostio.hip): every byte of caller memory that goes to or comes from the device passes through
it (into / out of the library's page-locked buffers).  No GPU needed."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

from positionbaseddynamics_amd import _ffi


def _copy(dst, dst_off, src, n):
    _ffi.lib.pbdx_debug_host_copy(C.c_void_p(dst.ctypes.data + dst_off), C.c_void_p(src.ctypes.data), n)


@pytest.mark.parametrize("n", [1, 100, (1 << 20) - 1, 1 << 20, (1 << 20) + 1, (12 << 20) + 13, (40 << 20) + 777])
def test_host_copy_copies_exactly_the_bytes_given(n):
    """Sizes below and above the threshold of the thread team, unaligned destination, guard bytes on both sides."""
    rng = np.random.default_rng(n & 0xffff)
    src = rng.integers(0, 256, n, dtype=np.uint8)
    dst = np.full(n + 64, 0xa5, dtype=np.uint8)
    _copy(dst, 7, src, n)
    assert np.array_equal(dst[7:7 + n], src)
    assert (dst[:7] == 0xa5).all() and (dst[7 + n:] == 0xa5).all()


def test_host_copy_from_several_threads_at_once():
    """The team takes one job at a time; callers on other threads (a plug-in's helper thread, the shards of a single-process ensemble) queue."""
    n = (6 << 20) + 5
    rng = np.random.default_rng(3)
    srcs = [rng.integers(0, 256, n, dtype=np.uint8) for _ in range(4)]
    dsts = [np.zeros(n, dtype=np.uint8) for _ in range(4)]

    def job(k):
        for _ in range(10):
            dsts[k][:] = 0
            _copy(dsts[k], 0, srcs[k], n)

    threads = [threading.Thread(target=job, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for d, s in zip(dsts, srcs):
        assert np.array_equal(d, s)


def test_host_copy_in_a_forked_child():
    """The child of a fork() has no worker threads: it must copy on its own thread instead of waiting for them."""
    n = (8 << 20) + 3
    src = np.arange(n, dtype=np.uint32).view(np.uint8)[:n].copy()
    warm = np.zeros(n, dtype=np.uint8)
    _copy(warm, 0, src, n)                    # (the team exists in the parent now)
    pid = os.fork()
    if pid == 0:
        code = 3
        try:
            dst = np.zeros(n, dtype=np.uint8)
            _copy(dst, 0, src, n)
            code = 0 if np.array_equal(dst, src) else 4
        finally:
            os._exit(code)
    _, status = os.waitpid(pid, 0)
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0
