"""how long hipMalloc / hipFree / hipHostMalloc take on this box, by size (the arena's chunk policy and the pinned buffers go by this)"""
import ctypes as C, time
hip = C.CDLL("libamdhip64.so")
def t(f):
    a = time.perf_counter(); f(); return (time.perf_counter() - a) * 1e3
p = C.c_void_p()
hip.hipSetDevice(0)
hip.hipMalloc(C.byref(p), C.c_size_t(1 << 20)); hip.hipFree(p)
for mb in (64, 256, 1024, 2048, 4096):
    n = C.c_size_t(mb << 20)
    a = t(lambda: hip.hipMalloc(C.byref(p), n)); b = t(lambda: hip.hipFree(p))
    a2 = t(lambda: hip.hipMalloc(C.byref(p), n)); b2 = t(lambda: hip.hipFree(p))
    print(f"hipMalloc {mb:5d} MB: {a:8.2f} ms, hipFree {b:8.2f} ms; again {a2:8.2f} / {b2:8.2f} ms")
for mb in (16, 64, 256):
    n = C.c_size_t(mb << 20)
    a = t(lambda: hip.hipHostMalloc(C.byref(p), n, 0)); b = t(lambda: hip.hipHostFree(p))
    print(f"hipHostMalloc {mb:5d} MB: {a:8.2f} ms, hipHostFree {b:8.2f} ms")
