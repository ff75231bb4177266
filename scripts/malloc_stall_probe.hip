// What does a large hipMalloc on a helper thread cost the thread that drives kernels?  (the arena's spare chunk: a step of the
// bench took 16-20 ms longer while a 5.4 GB chunk was being allocated beside it)
//   build: hipcc --offload-arch=gfx950 -O2 -o /tmp/malloc_stall_probe scripts/malloc_stall_probe.hip -lpthread
//   run:   /tmp/malloc_stall_probe VARIANT GB [SKIP_GB]     VARIANT: 0 one hipMalloc, 1 pieces of 128 MB with 2 ms pauses,
//          2 hipMallocAsync from the default pool, 3 hipMemCreate + hipMemMap (virtual memory API), 4 no allocation (control)
//          SKIP_GB: allocated (and kept) before the measurement starts, to reach VRAM no process has touched since boot
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

__global__ void touch(uint32_t *p, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] += 1;
}

static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const int variant = argc > 1 ? atoi(argv[1]) : 0;
    const size_t gb = argc > 2 ? (size_t)atoi(argv[2]) : 6;
    const size_t skip = argc > 3 ? (size_t)atoi(argv[3]) : 0;
    hipSetDevice(0);
    hipStream_t st, st2;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&st2, hipStreamNonBlocking);
    void *skipped = nullptr;
    if (skip)
        hipMalloc(&skipped, skip << 30);
    uint32_t *buf = nullptr;
    const size_t n = (size_t)64 << 20;
    hipMalloc((void **)&buf, n * 4);
    hipLaunchKernelGGL(touch, dim3(2048), dim3(256), 0, st, buf, n);
    hipStreamSynchronize(st);

    std::atomic<int> phase{0}; // 1: allocation running, 2: done
    double alloc_ms = 0, t_a0 = 0, t_a1 = 0;
    std::thread helper([&] {
        std::this_thread::sleep_for(std::chrono::milliseconds(300));
        t_a0 = now();
        phase = 1;
        const size_t bytes = gb << 30;
        void *p = nullptr;
        if (variant == 0) {
            if (hipMalloc(&p, bytes) != hipSuccess)
                printf("hipMalloc failed\n");
        } else if (variant == 1) {
            for (size_t done = 0; done < bytes; done += (size_t)128 << 20) {
                hipMalloc(&p, (size_t)128 << 20);
                std::this_thread::sleep_for(std::chrono::milliseconds(2));
            }
        } else if (variant == 2) {
            if (hipMallocAsync(&p, bytes, st2) != hipSuccess)
                printf("hipMallocAsync failed\n");
            hipStreamSynchronize(st2);
        } else if (variant == 3) {
            hipMemAllocationProp prop = {};
            prop.type = hipMemAllocationTypePinned;
            prop.location.type = hipMemLocationTypeDevice;
            prop.location.id = 0;
            size_t gran = 0;
            hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
            hipMemGenericAllocationHandle_t h;
            hipDeviceptr_t va = nullptr;
            hipError_t e = hipMemAddressReserve(&va, bytes, gran, nullptr, 0);
            if (e == hipSuccess)
                e = hipMemCreate(&h, bytes, &prop, 0);
            if (e == hipSuccess)
                e = hipMemMap(va, bytes, 0, h, 0);
            hipMemAccessDesc acc = {};
            acc.location = prop.location;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            if (e == hipSuccess)
                e = hipMemSetAccess(va, bytes, &acc, 1);
            if (e != hipSuccess)
                printf("virtual memory path failed: %s\n", hipGetErrorString(e));
        }
        t_a1 = now();
        alloc_ms = t_a1 - t_a0;
        phase = 2;
    });

    struct It { double t, ms; int ph; };
    std::vector<It> its;
    const double t0 = now();
    while (now() - t0 < 1500) {
        const double a = now();
        const int ph = phase.load();
        hipLaunchKernelGGL(touch, dim3(2048), dim3(256), 0, st, buf, n);
        hipStreamSynchronize(st);
        its.push_back({a - t0, now() - a, ph});
    }
    helper.join();
    double med = 0;
    {
        std::vector<double> v;
        for (auto &i : its)
            v.push_back(i.ms);
        std::nth_element(v.begin(), v.begin() + v.size() / 2, v.end());
        med = v[v.size() / 2];
    }
    double worst = 0, lost = 0;
    int slow = 0;
    for (auto &i : its)
        if (i.t + i.ms >= t_a0 - t0 - 1 && i.t <= t_a1 - t0 + 20) {
            worst = std::max(worst, i.ms);
            if (i.ms > 2 * med) {
                lost += i.ms - med;
                ++slow;
            }
        }
    printf("variant %d, %zu GB (after %zu GB kept): allocation %.1f ms; kernel+wait median %.3f ms, worst beside the allocation %.2f ms, "
           "%d slow iterations, %.1f ms lost\n", variant, gb, skip, alloc_ms, med, worst, slow, lost);
    return 0;
}
