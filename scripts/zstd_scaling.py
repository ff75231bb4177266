import ctypes as C, time, threading, sys, os
z = C.CDLL("/opt/conda/lib/libzstd.so.1")
z.ZSTD_createCCtx.restype = C.c_void_p
z.ZSTD_compressCCtx.restype = C.c_size_t
z.ZSTD_compressCCtx.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
z.ZSTD_compressBound.restype = C.c_size_t
z.ZSTD_compressBound.argtypes = [C.c_size_t]
import random
random.seed(1)
def delta():
    out = []
    for _ in range(60):
        out.append(f"{random.choice('ABCD')}0,{random.randint(10,3000)}.")
    return "".join(out).encode()
pack = b"".join(delta() + b"\xff" for _ in range(6))
print("pack", len(pack), "cpus", os.cpu_count())
def worker(n, res, i):
    cctx = z.ZSTD_createCCtx()
    dst = C.create_string_buffer(z.ZSTD_compressBound(len(pack)))
    t0 = time.time()
    for _ in range(n):
        z.ZSTD_compressCCtx(cctx, dst, len(dst), pack, len(pack), 17)
    res[i] = time.time() - t0
for nt in (1, 16, 64, 128, 256):
    n = 400
    res = [0] * nt
    th = [threading.Thread(target=worker, args=(n, res, i)) for i in range(nt)]
    t0 = time.time()
    [t.start() for t in th]; [t.join() for t in th]
    wall = time.time() - t0
    print(f"threads {nt:3d}: {n*nt} calls in {wall:.3f} s -> {n*nt/wall:9.0f} calls/s, per-thread avg {sum(res)/nt/n*1e3:.3f} ms/call")
