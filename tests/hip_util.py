"""Device buffers for tests that hand raw device pointers to the C ABI: hipMalloc / hipMemcpy / hipFree of the HIP runtime
libcrt.so is already bound to (same soname, so the same copy).  Test plumbing only."""
import ctypes as C


class Hip:
    """hipMalloc / hipMemcpy / hipFree of the HIP runtime libcrt.so is already bound to (same soname, same copy): the
    caller's device buffers.  Test plumbing only."""

    def __init__(self):
        self.lib = C.CDLL("libamdhip64.so.7")
        self.lib.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.lib.hipFree.argtypes = [C.c_void_p]
        self.lib.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.lib.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.lib.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]

    def malloc(self, n):
        p = C.c_void_p()
        assert self.lib.hipMalloc(C.byref(p), n) == 0
        return p.value

    def free(self, p):
        assert self.lib.hipFree(C.c_void_p(p)) == 0

    def h2d(self, dst, a):
        assert self.lib.hipMemcpy(C.c_void_p(dst), a.ctypes.data_as(C.c_void_p), a.nbytes, 1) == 0

    def d2h(self, a, src):
        assert self.lib.hipMemcpy(a.ctypes.data_as(C.c_void_p), C.c_void_p(src), a.nbytes, 2) == 0

    def mem_free(self):
        f, t = C.c_size_t(), C.c_size_t()
        assert self.lib.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
        return f.value

    def memset(self, p, v, n):
        assert self.lib.hipMemset(C.c_void_p(p), v, n) == 0

    # --- importing a dma-buf descriptor the way another API would (tests of crt_image_export_fd) ---
    class _Loc(C.Structure):
        _fields_ = [("type", C.c_int), ("id", C.c_int)]

    class _Access(C.Structure):
        _fields_ = [("type", C.c_int), ("id", C.c_int), ("flags", C.c_int)]

    def import_fd(self, fd, n_bytes, device=0):
        """hipMemImportFromShareableHandle + reserve + map + access; returns (device address, handle)."""
        L = self.lib
        L.hipMemImportFromShareableHandle.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int]
        L.hipMemAddressReserve.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_size_t, C.c_void_p, C.c_ulonglong]
        L.hipMemMap.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_ulonglong]
        L.hipMemSetAccess.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(self._Access), C.c_size_t]
        h, va = C.c_void_p(), C.c_void_p()
        rc = L.hipMemImportFromShareableHandle(C.byref(h), C.c_void_p(fd), 1)
        assert rc == 0, f"hipMemImportFromShareableHandle: {rc}"
        rc = L.hipMemAddressReserve(C.byref(va), n_bytes, 0, None, 0)
        assert rc == 0, f"hipMemAddressReserve: {rc}"
        rc = L.hipMemMap(va, n_bytes, 0, h, 0)
        assert rc == 0, f"hipMemMap: {rc}"
        acc = self._Access(1, device, 3)
        rc = L.hipMemSetAccess(va, n_bytes, C.byref(acc), 1)
        assert rc == 0, f"hipMemSetAccess: {rc}"
        return va.value, h.value

    def release_import(self, va, h, n_bytes):
        L = self.lib
        L.hipMemUnmap.argtypes = [C.c_void_p, C.c_size_t]
        L.hipMemAddressFree.argtypes = [C.c_void_p, C.c_size_t]
        L.hipMemRelease.argtypes = [C.c_void_p]
        assert L.hipMemUnmap(C.c_void_p(va), n_bytes) == 0
        assert L.hipMemAddressFree(C.c_void_p(va), n_bytes) == 0
        assert L.hipMemRelease(C.c_void_p(h)) == 0
