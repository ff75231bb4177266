// zstd_kernels.hip -- S3 on the GPU: ZSTD_compressCCtx(level 17) frames for delta packs (src/common/segment.h:258-280 ->
// :199-201), byte-identical to libzstd 1.4.9 (restated in agc_amd/csrc/zstd/*.h; parity: tests/test_zstd_frames.py on the
// host build of the same headers, tests/test_gpu_zstd.py through this kernel).
//
// Mapping: the greedy/optimal parse of one frame is a serial dependency chain (binary-tree insertions, price table), but a
// collection closes tens of thousands of INDEPENDENT packs at once (one per group): one frame per LANE, every table of a
// frame in its own slice of an HBM arena (the three small frequency tables of the price model in LDS).  No MFMA: integer/byte
// work bound by dependent memory latency; the parallelism is the number of frames in flight.
#include "dev_common.h"
#include "zstd/zs_frame.h"
#include "zstd/zs_params.h"

namespace agc {

// (offsets from the three base pointers the kernel receives as ARGUMENTS: pointers loaded from memory would be generic
// ("flat") to the compiler -- slower instructions and, worse, waits on every outstanding access before each use)
struct ZFrameJob {
    uint64_t src;       // offset of the input in the source buffer
    uint64_t dst;       // offset of the frame slot (zs::frameBound(src_size) bytes) in the output buffer
    uint64_t ws;        // offset of the workspace (zs::wsLayout(cp, src_size).total bytes, zeroed) in the arena
    uint32_t src_size;
    uint32_t idx;       // index in the caller's order
    zs::CParams cp;
    uint32_t pad;
};

// lanes_per_wave < 64 leaves lanes idle on purpose: fewer frames per wave = less control-flow divergence inside a wave and
// more waves per SIMD to hide memory latency behind each other (the frames of one call rarely fill the chip's wave slots)
// WPS = waves per SIMD the register allocation leaves room for: with few lanes per wave a call has several waves per SIMD,
// and their memory waits overlap only if they are resident together
// USE_LDS = false: the tables stay in the workspace -- a launch that runs for a second beside the steps of the next samples must
// not take the LDS the 128 KiB splitter-scan blocks need (with it a scan block waits until a CU has drained its frames)
template <int WPS, bool USE_LDS, bool M_LDS = false>
__global__ void __launch_bounds__(64, WPS) zstd_frames_kernel(const ZFrameJob *__restrict__ jobs, uint32_t n_jobs, uint32_t *__restrict__ out_size,
                                                         uint32_t lanes_per_wave, const uint8_t *__restrict__ src_base, uint8_t *__restrict__ dst_base,
                                                         uint8_t *__restrict__ ws_base, uint32_t debug)
{
    if (threadIdx.x >= lanes_per_wave)
        return;
    const uint32_t j = blockIdx.x * lanes_per_wave + threadIdx.x;
    if (j >= n_jobs)
        return;
    const ZFrameJob jb = jobs[j];
    // the three small frequency tables of the price model (121 words per frame) live in LDS: the price loops look them up for
    // every candidate length; an odd stride keeps the lanes of a wave on different banks for equal indices
    extern __shared__ uint32_t zs_lds[];
    uint32_t *const fast = zs_lds + threadIdx.x * (M_LDS ? zs::FAST_WORDS : zs::FAST_FREQ_WORDS);
    out_size[jb.idx] = zs::compressFrame<USE_LDS, M_LDS>(ws_base + jb.ws, jb.cp, src_base + jb.src, jb.src_size, dst_base + jb.dst, false, debug, fast);
}

template __global__ void zstd_frames_kernel<2, false>(const ZFrameJob *, uint32_t, uint32_t *, uint32_t, const uint8_t *, uint8_t *, uint8_t *, uint32_t);
// (the default: frequency tables and the first matches of a request in LDS)
template __global__ void zstd_frames_kernel<2, true, true>(const ZFrameJob *, uint32_t, uint32_t *, uint32_t, const uint8_t *, uint8_t *, uint8_t *, uint32_t);


// ---- several lanes per frame (zstd/zs_opt_grp.h): G consecutive lanes = one group = one frame ----
// LDS per wave: per group the exchange record + the three small frequency tables, per lane the record of a tree walk.
constexpr uint32_t ZGRP_STRIDE = (zs::GRPX_WORDS + zs::FAST_FREQ_WORDS) | 1; // words per group (odd: neighbouring groups on different banks)
constexpr uint32_t ZGRP_REC_STRIDE = zs::GRP_RC | 1;                          // words per lane (inputs <= 16 KiB: one word per record)
constexpr uint32_t ZGRP_REC_STRIDE_WIDE = (2 * zs::GRP_RC) | 1;               // ... two words per record (zs::grpWide)
// (G = 3: 21 groups -> 20 328 bytes: EIGHT waves per CU (160 KiB), i.e. 43 008 frames of a launch resident at once)
__host__ __device__ static inline uint32_t zgrp_lds_bytes(uint32_t groups_per_wave, uint32_t g, bool wide = false)
{
    return (groups_per_wave * ZGRP_STRIDE + groups_per_wave * g * (wide ? ZGRP_REC_STRIDE_WIDE : ZGRP_REC_STRIDE)) * 4;
}

template <int G, int WPS>
__global__ void __launch_bounds__(64, WPS) zstd_frames_grp_kernel(const ZFrameJob *__restrict__ jobs, uint32_t n_jobs, uint32_t *__restrict__ out_size,
                                                                 uint32_t groups_per_wave, const uint8_t *__restrict__ src_base,
                                                                 uint8_t *__restrict__ dst_base, uint8_t *__restrict__ ws_base, uint32_t debug,
                                                                 uint32_t rec_stride)
{
    const uint32_t grp = threadIdx.x / G, j = threadIdx.x % G;
    if (grp >= groups_per_wave)
        return;
    const uint32_t job = blockIdx.x * groups_per_wave + grp;
    if (job >= n_jobs)
        return;
    extern __shared__ uint32_t zs_lds[];
    uint32_t *const gbase = zs_lds + grp * ZGRP_STRIDE;
    zs::GrpX &sh = *(zs::GrpX *)gbase;
    zs::GLane l;
    l.j = j;
    l.recs = (zs::ZS_LDS_U32P)(zs_lds + groups_per_wave * ZGRP_STRIDE + threadIdx.x * rec_stride);
    const ZFrameJob jb = jobs[job];
    const uint32_t n = zs::compressFrameGrp<G>(&l, sh, ws_base + jb.ws, jb.cp, src_base + jb.src, jb.src_size, dst_base + jb.dst, debug, gbase + zs::GRPX_WORDS);
    if (j == 0)
        out_size[jb.idx] = n;
}
template __global__ void zstd_frames_grp_kernel<2, 2>(const ZFrameJob *, uint32_t, uint32_t *, uint32_t, const uint8_t *, uint8_t *, uint8_t *, uint32_t, uint32_t);
template __global__ void zstd_frames_grp_kernel<3, 2>(const ZFrameJob *, uint32_t, uint32_t *, uint32_t, const uint8_t *, uint8_t *, uint8_t *, uint32_t, uint32_t);

// frames (scattered, padded slots) -> one contiguous buffer in the caller's order
__global__ void __launch_bounds__(256) zstd_gather_kernel(const ZFrameJob *__restrict__ jobs, uint32_t n_jobs, const uint64_t *__restrict__ dst_off,
                                                          const uint8_t *__restrict__ dst_base, uint8_t *__restrict__ out)
{
    for (uint32_t j = blockIdx.x; j < n_jobs; j += gridDim.x) {
        const ZFrameJob jb = jobs[j];
        const uint64_t o = dst_off[jb.idx];
        const uint32_t len = (uint32_t)(dst_off[jb.idx + 1] - o);
        for (uint32_t t = threadIdx.x; t < len; t += blockDim.x)
            out[o + t] = dst_base[jb.dst + t];
    }
}

} // namespace agc
