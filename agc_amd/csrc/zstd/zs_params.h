// Derived from Zstandard 1.4.9 (Copyright (c) 2016-present, Facebook, Inc.; BSD license): see NOTICE in this directory.
// zs_params.h -- ZSTD_getCParams(level, srcSize, 0) of libzstd 1.4.9 for a KNOWN source size and the three levels agc uses
// (segment.h:172-255: 17 for delta packs, 13 for tuple-packed references, 19 for repetitive references): the rows of
// ZSTD_defaultCParameters (one per source-size class) followed by ZSTD_adjustCParams_internal (zstd_compress.c).
// Host code (the C ABI decides the parameters, the kernels receive them); pinned against the library by
// tests/test_zstd_frames.py::test_level_parameters_equal_getcparams.
#pragma once
#include "zs_common.h"

namespace zs {

enum { STRAT_BTLAZY2_ = 6 }; // (only named by the level-13 row of the largest class, which no entry point takes)

// level: 13, 17 or 19 (anything else: 17)
ZHD void levelParams(int level, uint64_t src_size, uint32_t p[7])
{
    //                                     W   C   H   S mml  TL  strategy
    static const uint32_t R13[4][7] = {{14, 15, 14, 5, 3, 32, STRAT_BTULTRA},   // <= 16 KB
                                       {17, 18, 17, 3, 4, 12, STRAT_BTOPT},     // <= 128 KB
                                       {18, 18, 19, 4, 4, 16, STRAT_BTOPT},     // <= 256 KB
                                       {22, 21, 22, 5, 5, 32, STRAT_BTLAZY2_}};
    static const uint32_t R17[4][7] = {{14, 15, 15, 6, 3, 128, STRAT_BTULTRA2},
                                       {17, 18, 17, 8, 3, 256, STRAT_BTULTRA},
                                       {18, 19, 19, 8, 3, 256, STRAT_BTULTRA},
                                       {23, 23, 22, 5, 4, 64, STRAT_BTOPT}};
    static const uint32_t R19[4][7] = {{14, 15, 15, 8, 3, 256, STRAT_BTULTRA2},
                                       {17, 18, 17, 5, 3, 256, STRAT_BTULTRA2},
                                       {18, 19, 19, 8, 3, 256, STRAT_BTULTRA2},
                                       {23, 24, 22, 7, 3, 256, STRAT_BTULTRA2}};
    const int cls = src_size <= 16 * 1024 ? 0 : src_size <= 128 * 1024 ? 1 : src_size <= 256 * 1024 ? 2 : 3;
    const uint32_t *r = level == 13 ? R13[cls] : level == 19 ? R19[cls] : R17[cls];
    for (int i = 0; i < 7; ++i)
        p[i] = r[i];
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

ZHD void level17Params(uint64_t src_size, uint32_t p[7]) { levelParams(17, src_size, p); }

} // namespace zs
