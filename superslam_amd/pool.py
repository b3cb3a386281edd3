"""DeviceDescriptors / DescriptorPool mirrors (include/DescriptorPool.h:13-91)."""
from __future__ import annotations

import ctypes as C

from . import _lib


class _SlotRef:
    """shared_ptr<void> slot_ref: returns the slot to the pool when the last copy dies.

    Holds its own reference on the pool's bookkeeping (sship_pool_retain), so it may run after the extractor that
    owns the pool was closed - in any order at interpreter exit - like the reference's deleter, which captures the
    shared FreeList and not the pool (include/DescriptorPool.h:71-75)."""

    def __init__(self, pool, slot):
        self.pool, self.slot = pool, slot
        _lib.lib().sship_pool_retain(pool)

    def __del__(self):
        try:
            if self.pool is not None and self.slot >= 0:
                L = _lib.lib()
                L.sship_pool_release(self.pool, self.slot)
                L.sship_pool_release_ref(self.pool)
                self.pool = None
        except Exception:
            pass


class DeviceDescriptors:
    """Device pointer + shape of descriptors resident in a pool slot ([count, dim] fp16)."""

    def __init__(self, data: int = 0, count: int = 0, dim: int = 0, slot: int = -1, pool=None):
        self.data, self.count, self.dim, self.slot = int(data or 0), count, dim, slot
        self.slot_ref = _SlotRef(pool, slot) if (pool is not None and slot >= 0) else None

    def empty(self) -> bool:
        return self.data == 0 or self.count == 0


class DescriptorPool:
    """Fixed pool of N device slots of max_keypoints*dim fp16 (DescriptorPool.h:51-91)."""

    def __init__(self, num_slots: int, max_keypoints: int, dim: int):
        if not _lib._inited:
            _lib.init()
        self._h = C.c_void_p()
        _lib.check(_lib.lib().sship_pool_create(num_slots, max_keypoints, dim, C.byref(self._h)))
        self._dim, self._max_kp = dim, max_keypoints

    def close(self) -> None:
        """Free the device slots (DescriptorPool.cc:27-32); handles made from this pool stay safe to drop."""
        if self._h:
            _lib.lib().sship_pool_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def make(self, count: int) -> DeviceDescriptors:
        slot = _lib.lib().sship_pool_acquire(self._h)
        if slot < 0:
            return DeviceDescriptors(0, count, self._dim, -1)  # exhausted
        return DeviceDescriptors(_lib.lib().sship_pool_slot_ptr(self._h, slot), count, self._dim, slot, self._h)

    def slot_ptr(self, slot: int) -> int:
        return _lib.lib().sship_pool_slot_ptr(self._h, slot) or 0

    def in_use(self) -> int:
        return _lib.lib().sship_pool_in_use(self._h)

    def dim(self) -> int:
        return self._dim

    def max_keypoints(self) -> int:
        return self._max_kp
