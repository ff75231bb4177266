// zstd_sim.cpp -- TEST INFRASTRUCTURE (part of the CPU device stand-in, see agc_hip_sim.c): the S3 entry points of
// include/agc_hip.h on the host build of the encoder headers (agc_amd/csrc/zstd/*.h -- the very code the HIP kernel runs), so
// that the host pipeline's use of the device entropy stage is checked against the golden archives without a GPU.
#include "../../include/agc_hip.h"
#include "../../agc_amd/csrc/zstd/zs_frame.h"
#include "../../agc_amd/csrc/zstd/zs_params.h"
#include <vector>

extern "C" {

uint32_t agc_hip_zstd17_max_input(void) { return zs::BLOCKSIZE_MAX; }
uint32_t agc_hip_zstd17_resident_frames(agc_hip_ctx *) { return 64; }
int agc_hip_zstd17_batch(agc_hip_ctx *ctx, uint32_t n, const uint8_t *h_src, const uint64_t *h_src_off, uint8_t *h_dst, uint64_t dst_cap,
                         uint64_t *h_dst_off);
int agc_hip_zstd17_batch_dev(agc_hip_ctx *ctx, uint32_t n, const uint8_t *d_src, const uint64_t *h_src_off, uint8_t *h_dst, uint64_t dst_cap,
                             uint64_t *h_dst_off)
{
    return agc_hip_zstd17_batch(ctx, n, d_src, h_src_off, h_dst, dst_cap, h_dst_off); // (the stand-in's "HBM" is host memory)
} // (small on purpose: the host's split rule is exercised)

int agc_hip_zstd17_cparams(uint64_t src_size, uint32_t out7[7])
{
    if (!out7)
        return AGC_HIP_EINVAL;
    zs::level17Params(src_size, out7);
    return AGC_HIP_OK;
}

int agc_hip_zstd_cparams(int level, uint64_t src_size, uint32_t out7[7])
{
    if (!out7 || (level != 13 && level != 17 && level != 19))
        return AGC_HIP_EINVAL;
    zs::levelParams(level, src_size, out7);
    return AGC_HIP_OK;
}

int agc_hip_zstd17_batch(agc_hip_ctx *ctx, uint32_t n, const uint8_t *h_src, const uint64_t *h_src_off, uint8_t *h_dst, uint64_t dst_cap,
                         uint64_t *h_dst_off)
{
    return agc_hip_zstd_batch(ctx, n, h_src, h_src_off, nullptr, h_dst, dst_cap, h_dst_off);
}

int agc_hip_zstd_batch(agc_hip_ctx *ctx, uint32_t n, const uint8_t *h_src, const uint64_t *h_src_off, const uint8_t *h_level, uint8_t *h_dst,
                       uint64_t dst_cap, uint64_t *h_dst_off)
{
    if (!ctx || !h_dst_off || (n && !h_src_off))
        return AGC_HIP_EINVAL;
    h_dst_off[0] = 0;
    std::vector<std::vector<uint8_t>> frames(n);
    for (uint32_t i = 0; i < n; ++i) {
        const uint64_t len = h_src_off[i + 1] - h_src_off[i];
        if (len > zs::BLOCKSIZE_MAX)
            return AGC_HIP_EINVAL;
        const int level = h_level ? h_level[i] : 17;
        if (level != 13 && level != 17 && level != 19)
            return AGC_HIP_EINVAL;
        uint32_t p[7];
        zs::levelParams(level, len, p);
        const zs::CParams cp = {p[0], p[1], p[2], p[3], p[4], p[5], p[6]};
        std::vector<uint8_t> ws(zs::wsLayout(cp, (uint32_t)len).total, 0);
        frames[i].resize(zs::frameBound((uint32_t)len));
        frames[i].resize(zs::compressFrame(ws.data(), cp, h_src + h_src_off[i], (uint32_t)len, frames[i].data()));
        h_dst_off[i + 1] = h_dst_off[i] + frames[i].size();
    }
    if (h_dst_off[n] > dst_cap)
        return AGC_HIP_ECAP;
    for (uint32_t i = 0; i < n; ++i)
        memcpy(h_dst + h_dst_off[i], frames[i].data(), frames[i].size());
    return AGC_HIP_OK;
}
}
