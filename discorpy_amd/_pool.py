"""Recycling of host output arrays.

The reference returns a fresh array from every call.  For a 4096 x 4096 float32 frame the first touch
of 64 MiB of fresh pages costs more than the whole GPU round trip (4.4 ms of page faults against
2.4 ms for H2D + kernel + D2H, tools/time_host_breakdown.py), and a loop such as
``examples/example_05.py:62-65`` pays it on every iteration.  Outputs of the NumPy path are therefore
leased from a pool: the array handed to the caller owns its block like any other array, and when the
caller drops the last reference (including views) the block -- with its pages already faulted in --
goes back to the pool for the next call of the same size.  Nothing is ever reused while reachable.

``DISCORPY_AMD_HOST_POOL_MB`` caps the bytes kept idle (default 1024; 0 disables the pool); beyond the cap the
blocks that have been idle the longest are dropped.
"""
import collections
import os
import threading

import numpy as np

_MIN_BYTES = 1 << 20          # smaller outputs are not worth tracking
_PIN_MIN_BYTES = 16 << 20     # blocks from this size on are registered with the GPU (the library's direct-write path starts there)

# Registered (pinned, unswappable) host memory is bounded: a caller who keeps many large results alive -- 1800 projections, a centre
# grid -- must not pin tens of GB through thousands of registrations.  Blocks are registered only while the total stays below the
# cap (DISCORPY_AMD_PIN_MAX_MB, default: the pool's cap) and only where the library would actually take its direct-write path
# (dcp_get_option "host_direct_applies": the runtime cannot overlap an upload with a download, or host_direct = 2); beyond that a
# block is plain memory and the staged path serves it.
_pin_lock = threading.Lock()
_pinned_bytes = 0


def _pin_cap():
    v = os.environ.get("DISCORPY_AMD_PIN_MAX_MB")
    return int(float(v) * (1 << 20)) if v is not None else int(float(os.environ.get("DISCORPY_AMD_HOST_POOL_MB", "1024")) * (1 << 20))


def _may_pin(nbytes):
    """Reserve `nbytes` of the pinned budget (True) or refuse."""
    global _pinned_bytes
    if nbytes < _PIN_MIN_BYTES or os.environ.get("DISCORPY_AMD_PIN_OUTPUTS", "1") == "0":
        return False
    try:
        from . import _ffi as F
        if F.device_count() <= 0:
            return False
        if not F.get_option("host_direct_applies"):         # (two atomic loads and a once-per-process probe inside the library)
            return False
    except Exception:      # noqa: BLE001 -- no library / no device
        return False
    with _pin_lock:
        if _pinned_bytes + nbytes > _pin_cap():
            return False
        _pinned_bytes += nbytes
        return True


def _unpin(nbytes):
    global _pinned_bytes
    with _pin_lock:
        _pinned_bytes = max(0, _pinned_bytes - nbytes)


def pinned_bytes():
    return _pinned_bytes


class _Block:
    """One block of the pool: a uint8 array, registered with the GPU (hipHostRegister through dcp_host_register) when it is large
    enough for the library's direct-write path -- the kernels then write a host frame's result straight into it instead of
    staging it on the device and copying it back (csrc/api_image.cpp: run_host_direct).  The registration costs ~0.8 ms per 64 MiB
    once per block and is undone before the memory goes away.  DISCORPY_AMD_PIN_OUTPUTS=0 switches it off."""

    __slots__ = ("arr", "registered", "_map", "_span")

    def __init__(self, nbytes):
        self.registered = False
        self._map = None
        self._span = 0
        if _may_pin(nbytes):
            # A block that gets REGISTERED lives in an anonymous mapping of its own, whole pages, and the whole pages are what is
            # registered: a malloc'ed block (glibc serves up to 32 MiB from the brk heap once large arrays have been freed) shares
            # its first and last page with neighbouring heap objects, and when one of those neighbours is the pageable source or
            # destination of another copy the runtime pins and unpins that shared page on the fly under the registration --
            # round 4's large-frame campaign ended in "Memory access fault by GPU" on a heap address about once per 10 000 cases.
            try:
                import mmap
                from . import _ffi as F
                page = mmap.PAGESIZE
                self._span = (nbytes + page - 1) // page * page
                self._map = mmap.mmap(-1, self._span)
                self.arr = np.frombuffer(self._map, dtype=np.uint8, count=nbytes)
                self.arr[::page] = 0                      # fault the pages in before they are pinned
                dev = int(os.environ.get("DISCORPY_AMD_DEVICE", "-1"))
                self.registered = F.lib().dcp_host_register(self.arr.ctypes.data, self._span, dev) == 0
            except Exception:      # noqa: BLE001 -- a plain block
                self.registered = False
            if not self.registered:
                _unpin(nbytes)
                self._map = None
        if not self.registered:
            self.arr = np.empty(nbytes, np.uint8)

    @property
    def nbytes(self):
        return self.arr.nbytes

    def release(self):
        if self.registered:
            self.registered = False
            try:
                _unpin(self.arr.nbytes)
                from . import _ffi as F
                F.lib().dcp_host_unregister(self.arr.ctypes.data)
            except Exception:      # noqa: BLE001 -- interpreter shutdown
                pass

    def __del__(self):
        self.release()


class _Lease:
    """Owner of one block; exposes it through __array_interface__ and returns it to the pool when the
    last array referring to it is gone."""

    __slots__ = ("_pool", "_block", "__array_interface__")

    def __init__(self, pool, block, shape, dtype):
        self._pool = pool
        self._block = block
        self.__array_interface__ = {"version": 3, "shape": tuple(int(s) for s in shape), "typestr": np.dtype(dtype).str,
                                    "data": (block.arr.ctypes.data, False), "strides": None}

    def __del__(self):
        pool, block = self._pool, self._block
        self._block = None
        if pool is not None and block is not None:
            pool._give_back(block)


class HostPool:
    def __init__(self, cap_bytes):
        self.cap = int(cap_bytes)
        self.idle = {}            # nbytes -> [uint8 blocks], most recently returned last
        self.age = collections.deque()   # nbytes of the idle blocks, oldest return first (eviction order)
        self.idle_bytes = 0
        self.lock = threading.Lock()
        self.hits = self.misses = 0

    def empty(self, shape, dtype):
        dtype = np.dtype(dtype)
        shape = tuple(int(s) for s in shape)
        nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        if self.cap <= 0 or nbytes < _MIN_BYTES:
            return np.empty(shape, dtype)
        block = None
        with self.lock:
            stack = self.idle.get(nbytes)
            if stack:
                block = stack.pop()
                self.idle_bytes -= nbytes
                self.age.remove(nbytes)
                self.hits += 1
            else:
                self.misses += 1
        if block is not None and not block.registered and nbytes >= _PIN_MIN_BYTES:
            # a block that was created plain -- the pinned budget was used up, or the direct-write path did not apply at the time
            # (host_direct set later) -- gets another chance when it is handed out again: a registered twin replaces it (a plain
            # block cannot be registered in place: it shares its edge pages with the heap).  ADVICE r4
            if _pinned_bytes + nbytes <= _pin_cap() and os.environ.get("DISCORPY_AMD_PIN_OUTPUTS", "1") != "0":
                twin = _Block(nbytes)
                if twin.registered:
                    block = twin
        if block is None:
            block = _Block(nbytes)
        return np.asarray(_Lease(self, block, shape, dtype))

    def _give_back(self, block):
        try:
            dropped = []
            with self.lock:
                if block.nbytes > self.cap:
                    dropped.append(block)
                else:
                    self.idle.setdefault(block.nbytes, []).append(block)
                    self.age.append(block.nbytes)
                    self.idle_bytes += block.nbytes
                    while self.idle_bytes > self.cap:        # evict the blocks that have been idle the longest
                        old = self.age.popleft()
                        dropped.append(self.idle[old].pop(0))
                        self.idle_bytes -= old
            for b in dropped:                                # (unregistered before their memory is freed)
                b.release()
        except Exception:      # interpreter shutdown: let the block go
            pass

    def clear(self):
        with self.lock:
            blocks = [b for stack in self.idle.values() for b in stack]
            self.idle.clear()
            self.age.clear()
            self.idle_bytes = 0
        for b in blocks:
            b.release()


_pool = HostPool(int(float(os.environ.get("DISCORPY_AMD_HOST_POOL_MB", "1024")) * (1 << 20)))


def empty(shape, dtype):
    """np.empty(shape, dtype) whose memory is recycled once the caller has dropped it."""
    return _pool.empty(shape, dtype)


def stats():
    return {"hits": _pool.hits, "misses": _pool.misses, "idle_bytes": _pool.idle_bytes, "cap_bytes": _pool.cap,
            "pinned_bytes": _pinned_bytes, "pin_cap_bytes": _pin_cap()}


def clear():
    _pool.clear()
