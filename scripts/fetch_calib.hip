// fetch_calib.hip -- calibration of rocprofv3's FETCH_SIZE on gfx950 for the access patterns of the LZ kernels.
//   hipcc --offload-arch=gfx950 -O3 scripts/fetch_calib.hip -o scripts/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-include-regex calib -d gpurun_out/calib -- scripts/fetch_calib
// Kernels (each touches a 4 GiB table, 16x the 256 MiB Infinity Cache, so re-use cannot hide requests):
//   calib_stream      every lane one aligned 16-B load, consecutive lanes consecutive addresses   (1 KiB per wave instruction)
//   calib_probe_row   the exact-step probe: 64 lanes read 64 consecutive 4-B slots at a random 4-B aligned row start
//   calib_probe16     the wide literal probe: every lane one 16-B load at its own random 4-B aligned address
//   calib_byte16      the match compare: every lane one 16-B load, lanes consecutive, base address byte-aligned (odd)
//   calib_probe4      the zstd match finder's pattern: every lane one 4-B load at its own random 64-B aligned address
//   calib_probe4_pair the same plus the 4 bytes 64 B further in the same 128-B aligned line: FETCH_SIZE per lane equal to
//                     calib_probe4's means the memory side moves 128-B lines (and the counter's unit is to be read accordingly);
//                     twice calib_probe4's means 64-B requests, counted as they are
// The program prints the bytes each kernel asks for; FETCH_SIZE (KB) of the same dispatch is the other half of the ratio.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

__global__ void calib_stream(const uint4 *t, uint64_t n16, uint32_t *sink)
{
    uint32_t acc = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 v = t[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

__global__ void calib_probe_row(const uint32_t *t, uint64_t n4, uint32_t iters, uint32_t *sink)
{
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        const uint64_t row = mix(((uint64_t)wave << 32) | it) % (n4 - 64);
        acc ^= t[row + lane];
    }
    if (acc == 0x12345678u) *sink = acc;
}

__global__ void calib_probe16(const uint8_t *t, uint64_t nbytes, uint32_t iters, uint32_t *sink)
{
    const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        const uint64_t a = (mix((tid << 20) | it) % ((nbytes - 16) >> 2)) << 2;
        uint4 v;
        __builtin_memcpy(&v, t + a, 16);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

__global__ void calib_byte16(const uint8_t *t, uint64_t nbytes, uint32_t *sink)
{
    uint32_t acc = 0;
    const uint64_t n16 = (nbytes - 64) / 16;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
        uint4 v;
        __builtin_memcpy(&v, t + 3 + i * 16, 16);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

__global__ void calib_probe4(const uint8_t *t, uint64_t nbytes, uint32_t iters, uint32_t pair, uint32_t *sink)
{
    const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        const uint64_t a = (mix((tid << 20) | it | 0x80000ull) % (nbytes >> 7)) << 7; // a 128-B aligned line of its own
        acc ^= *(const uint32_t *)(t + a);
        if (pair)
            acc ^= *(const uint32_t *)(t + a + 64);
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void calib_probe4_pair(const uint8_t *t, uint64_t nbytes, uint32_t iters, uint32_t pair, uint32_t *sink)
{
    const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        const uint64_t a = (mix((tid << 20) | it | 0x80000ull) % (nbytes >> 7)) << 7;
        acc ^= *(const uint32_t *)(t + a);
        if (pair)
            acc ^= *(const uint32_t *)(t + a + 64);
    }
    if (acc == 0x12345678u) *sink = acc;
}

int main()
{
    const uint64_t nbytes = 4ull << 30;
    uint8_t *t = nullptr;
    uint32_t *sink = nullptr;
    if (hipMalloc((void **)&t, nbytes) != hipSuccess || hipMalloc((void **)&sink, 4) != hipSuccess) {
        fprintf(stderr, "hipMalloc failed\n");
        return 1;
    }
    hipMemset(t, 0x5A, nbytes);
    hipDeviceSynchronize();
    const uint32_t grid = 2048, block = 256, iters = 256;
    hipLaunchKernelGGL(calib_stream, dim3(grid), dim3(block), 0, 0, (const uint4 *)t, nbytes / 16, sink);
    hipDeviceSynchronize();
    printf("calib_stream     asks for %llu bytes (16 B per lane, coalesced)\n", (unsigned long long)nbytes);
    hipLaunchKernelGGL(calib_probe_row, dim3(grid), dim3(block), 0, 0, (const uint32_t *)t, nbytes / 4, iters, sink);
    hipDeviceSynchronize();
    printf("calib_probe_row  asks for %llu bytes = %llu rows of 256 B (each row spans 2-3 128-B lines)\n",
           (unsigned long long)grid * block / 64 * iters * 256ull, (unsigned long long)grid * block / 64 * iters);
    hipLaunchKernelGGL(calib_probe16, dim3(grid), dim3(block), 0, 0, (const uint8_t *)t, nbytes, iters, sink);
    hipDeviceSynchronize();
    printf("calib_probe16    asks for %llu bytes = %llu loads of 16 B at random 4-B aligned addresses\n",
           (unsigned long long)grid * block * iters * 16ull, (unsigned long long)grid * block * iters);
    hipLaunchKernelGGL(calib_byte16, dim3(grid), dim3(block), 0, 0, (const uint8_t *)t, nbytes, sink);
    hipDeviceSynchronize();
    printf("calib_byte16     asks for %llu bytes (16 B per lane, coalesced, base + 3)\n", (unsigned long long)((nbytes - 64) / 16 * 16));
    hipLaunchKernelGGL(calib_probe4, dim3(grid), dim3(block), 0, 0, (const uint8_t *)t, nbytes, iters, 0u, sink);
    hipDeviceSynchronize();
    printf("calib_probe4     asks for %llu bytes = %llu loads of 4 B, each in a 128-B line of its own (4 GiB table)\n",
           (unsigned long long)grid * block * iters * 4ull, (unsigned long long)grid * block * iters);
    hipLaunchKernelGGL(calib_probe4_pair, dim3(grid), dim3(block), 0, 0, (const uint8_t *)t, nbytes, iters, 1u, sink);
    hipDeviceSynchronize();
    printf("calib_probe4_pair asks for %llu bytes = %llu pairs of 4-B loads 64 B apart in one 128-B line\n",
           (unsigned long long)grid * block * iters * 8ull, (unsigned long long)grid * block * iters);
    hipFree(t);
    hipFree(sink);
    return 0;
}
