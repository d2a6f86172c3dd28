"""Host link rates as the library's own copies see them: page-locked blocks of several sizes, host to device and back."""
import ctypes, json, sys, time
sys.path.insert(0, ".")
from rodio_amd import _lib

lib = _lib.load()
lib.rh_init(0)
st = ctypes.c_void_p()
lib.rh_stream_create(ctypes.byref(st))
res = []
for mb in (1, 4, 16, 64, 256):
    n = mb << 20
    h = ctypes.c_void_p(); d = ctypes.c_void_p()
    assert lib.rh_host_alloc(ctypes.byref(h), n) == 0
    assert lib.rh_malloc(ctypes.byref(d), n) == 0
    ctypes.memset(h, 1, n)
    for name, fn in (("h2d", lambda: lib.rh_memcpy_h2d(d, h, n, st)), ("d2h", lambda: lib.rh_memcpy_d2h_async(h, d, n, st))):
        fn(); lib.rh_stream_synchronize(st)
        reps = max(2, 512 // mb)
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        t_issue = time.perf_counter() - t0
        lib.rh_stream_synchronize(st)
        t = time.perf_counter() - t0
        res.append({"dir": name, "MiB": mb, "GBps": round(n * reps / t / 1e9, 2), "issue_frac": round(t_issue / t, 3)})
    lib.rh_free(d); lib.rh_host_free(h)
print(json.dumps(res))
