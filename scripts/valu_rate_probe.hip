// Issue rate of the integer instructions the scan's filter is made of (wave64 on a 16-lane SIMD: a full-rate instruction takes 4
// cycles; which ones take more?).  build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/vrp scripts/valu_rate_probe.hip && /tmp/vrp
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP> __global__ void __launch_bounds__(256) probe(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
        a[i] = seed + threadIdx.x * 8 + i;
    const uint32_t c = seed | 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0)
                a[i] = a[i] * c;                                                   // v_mul_lo_u32
            else if (OP == 1)
                a[i] = __builtin_amdgcn_alignbit(a[i], c, 7) ^ 0;                    // v_alignbit_b32 (the xor folds away)
            else if (OP == 2)
                a[i] = (a[i] & 0xFFFFFFu) * (c & 0xFFFFFFu) + (a[i] >> 20);           // v_lshrrev + v_mad_u32_u24
            else if (OP == 3)
                a[i] = a[i] ^ (a[i] >> 7);                                           // v_lshrrev + v_xor
            else if (OP == 4)
                a[i] = (a[i] & 0xFFFFFFu) * (c & 0xFFFFFFu);                          // v_mul_u32_u24
            else if (OP == 5)
                a[i] = __mulhi(a[i], c);                                             // v_mul_hi_u32
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP> static void run(const char *what, int ops_per_iter)
{
    uint32_t *d = nullptr;
    const int blocks = 256 * 8, iters = 20000;
    hipMalloc((void **)&d, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(256), 0, 0, d, 12345u, 100);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(256), 0, 0, d, 12345u, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions = blocks * 4 waves * iters * 8 * ops_per_iter; 1024 SIMDs
    const double winst = (double)blocks * 4 * iters * 8 * ops_per_iter;
    const double per_simd_per_us = winst / 1024.0 / (ms * 1e3);
    printf("%-44s %8.2f ms  %6.1f wave-instructions per SIMD and microsecond (%d per iteration)\n", what, ms, per_simd_per_us, ops_per_iter);
    hipFree(d);
}

int main()
{
    run<1>("v_alignbit_b32", 1);
    run<0>("v_mul_lo_u32", 1);
    run<4>("v_mul_u32_u24", 1);
    run<2>("v_lshrrev + v_mad_u32_u24", 2);
    run<3>("v_lshrrev + v_xor", 2);
    run<5>("v_mul_hi_u32", 1);
    return 0;
}
