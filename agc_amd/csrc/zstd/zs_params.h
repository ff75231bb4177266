// Derived from Zstandard 1.4.9 (Copyright (c) 2016-present, Facebook, Inc.; BSD license): see NOTICE in this directory.
// zs_params.h -- ZSTD_getCParams(17, srcSize, 0) of libzstd 1.4.9 for a KNOWN source size: the level-17 rows of
// ZSTD_defaultCParameters (one per source-size class) followed by ZSTD_adjustCParams_internal (zstd_compress.c).
// Host code (the C ABI decides the parameters, the kernels receive them); pinned against the library by
// tests/test_zstd_frames.py::test_level17_parameters_equal_getcparams.
#pragma once
#include "zs_common.h"

namespace zs {

ZHD void level17Params(uint64_t src_size, uint32_t p[7])
{
    //              W   C   H   S mml  TL  strategy
    if (src_size <= 16 * 1024) {
        const uint32_t r[7] = {14, 15, 15, 6, 3, 128, STRAT_BTULTRA2};
        for (int i = 0; i < 7; ++i) p[i] = r[i];
    } else if (src_size <= 128 * 1024) {
        const uint32_t r[7] = {17, 18, 17, 8, 3, 256, STRAT_BTULTRA};
        for (int i = 0; i < 7; ++i) p[i] = r[i];
    } else if (src_size <= 256 * 1024) {
        const uint32_t r[7] = {18, 19, 19, 8, 3, 256, STRAT_BTULTRA};
        for (int i = 0; i < 7; ++i) p[i] = r[i];
    } else {
        const uint32_t r[7] = {23, 23, 22, 5, 4, 64, STRAT_BTOPT};
        for (int i = 0; i < 7; ++i) p[i] = r[i];
    }
    if (src_size < (1ULL << 30)) { // resize windowLog if the input is small enough
        const uint32_t tSize = (uint32_t)src_size;
        const uint32_t srcLog = (tSize < (1u << 6)) ? 6 : highbit32(tSize - 1) + 1;
        if (p[0] > srcLog)
            p[0] = srcLog;
    }
    {
        const uint32_t cycleLog = p[1] - 1; // bt strategies: chainLog - 1
        if (p[2] > p[0] + 1)
            p[2] = p[0] + 1;
        if (cycleLog > p[0])
            p[1] -= cycleLog - p[0];
    }
    if (p[0] < 10)
        p[0] = 10; // ZSTD_WINDOWLOG_ABSOLUTEMIN
}

} // namespace zs
