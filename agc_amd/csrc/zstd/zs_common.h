// Derived from Zstandard 1.4.9 (Copyright (c) 2016-present, Facebook, Inc.; BSD license): see NOTICE in this directory.
// zs_common.h -- shared definitions of the level-17 zstd frame encoder (S3, SURVEY 8b: ZSTD_compressCCtx at
// src/common/segment.h:176,201) written for gfx950: one frame per lane, every table in a per-frame workspace in HBM.
//
// The code under agc_amd/csrc/zstd/ restates what libzstd 1.4.9 does for ZSTD_compressCCtx(level 17) on inputs of at most
// one block (<= 128 KiB): zstd's format is public (RFC 8878) but the BYTES depend on the encoder's match finder, price
// model and table heuristics, so those are followed decision by decision (zstd v1.4.9: lib/compress/zstd_opt.c,
// zstd_compress.c, zstd_compress_sequences.c, zstd_compress_literals.c, huf_compress.c, fse_compress.c, hist.c).  The
// reference tree has no copy of zstd (empty submodule): parity is pinned against the image's libzstd 1.4.9 itself --
// tests/test_zstd_frames.py compares every frame byte for byte, and the parser's sequences with ZSTD_generateSequences.
//
// The same headers compile for the host (gcc, tests only: tests/zstd_host) and for the device (hipcc, the product).
#pragma once
#include <stdint.h>
#include <string.h>

#ifdef __HIPCC__
// (always_inline: an outlined function takes generic pointers -- flat_* instead of global_*/ds_* accesses -- and a 128-VGPR budget)
#define ZFN __device__ static inline __attribute__((always_inline))
#define ZHD __host__ __device__ static inline   // also needed by the host side of the C ABI (sizes, parameters)
#define ZCONST __device__ static const
#else
#define ZFN static inline
#define ZHD static inline
#define ZCONST static const
#endif

namespace zs {

typedef uint8_t BYTE;
typedef uint16_t U16;
typedef uint32_t U32;
typedef uint64_t U64;

constexpr U32 OPT_NUM = 1u << 12;      // ZSTD_OPT_NUM
constexpr U32 REP_NUM = 3;             // ZSTD_REP_NUM
constexpr U32 REP_MOVE = 2;            // ZSTD_REP_MOVE
constexpr U32 MINMATCH = 3;
constexpr U32 MaxLit = 255, MaxLL = 35, MaxML = 52, MaxOff = 31;
constexpr U32 BLOCKSIZE_MAX = 1u << 17;
constexpr U32 HASHLOG3_MAX = 17;

enum { STRAT_BTOPT = 7, STRAT_BTULTRA = 8, STRAT_BTULTRA2 = 9 };

struct CParams {
    U32 windowLog, chainLog, hashLog, searchLog, minMatch, targetLength, strategy;
};

struct Match {
    U32 off, len;
};

struct alignas(16) Optimal { // ZSTD_optimal_t (padded to 32 bytes: two 16-byte accesses per entry)
    int price;
    U32 off, mlen, litlen;
    U32 rep[3];
    U32 pad;
};

struct Seq { // one stored sequence (full lengths; offCode = repcode 0..2 or distance + 2)
    U32 offCode, litLength, matchLength;
};

ZHD U32 highbit32(U32 v) { return 31u - (U32)__builtin_clz(v); }

ZFN U32 read32(const BYTE *p)
{
    U32 v;
    memcpy(&v, p, 4);
    return v;
}
ZFN U64 read64(const BYTE *p)
{
    U64 v;
    memcpy(&v, p, 8);
    return v;
}

// ZSTD_count: common length of ip / match, ip bounded by iend
ZFN U32 count(const BYTE *ip, const BYTE *match, const BYTE *iend)
{
    const BYTE *const start = ip;
    while (ip + 8 <= iend) {
        const U64 d = read64(ip) ^ read64(match);
        if (d)
            return (U32)(ip - start) + ((U32)__builtin_ctzll(d) >> 3);
        ip += 8;
        match += 8;
    }
    while (ip < iend && *ip == *match) {
        ++ip;
        ++match;
    }
    return (U32)(ip - start);
}

// ZSTD_count that also hands back the two bytes at the first difference (what the tree walk compares next): they are in the
// words just compared, a second trip to memory for them is wasted.  *diff = false when ip reached iend.
ZFN U32 countEx(const BYTE *ip, const BYTE *match, const BYTE *iend, U32 *ipByte, U32 *matchByte, bool *diff)
{
    const BYTE *const start = ip;
    while (ip + 8 <= iend) {
        const U64 a = read64(ip), b = read64(match);
        const U64 d = a ^ b;
        if (d) {
            const U32 sh = (U32)__builtin_ctzll(d) & ~7u;
            *ipByte = (U32)(a >> sh) & 0xFF;
            *matchByte = (U32)(b >> sh) & 0xFF;
            *diff = true;
            return (U32)(ip - start) + (sh >> 3);
        }
        ip += 8;
        match += 8;
    }
    while (ip < iend) {
        const U32 a = *ip, b = *match;
        if (a != b) {
            *ipByte = a;
            *matchByte = b;
            *diff = true;
            return (U32)(ip - start);
        }
        ++ip;
        ++match;
    }
    *diff = false;
    return (U32)(ip - start);
}

// Code tables of the zstd format (RFC 8878 3.1.1.3.2.1.1: literal-length and match-length codes and their extra bits) as
// ARITHMETIC: on the GPU a table lookup is a global-memory round trip per call, a few selects are not.  The tables the
// library uses (LL_Code / ML_Code / LL_bits / ML_bits, zstd_internal.h) are spelled out in tests/test_zstd_frames.py, which
// checks these functions against them for every argument.
ZHD U32 LLbits(U32 code) // LL_bits[code], code <= 35
{
    return code < 16 ? 0 : code < 20 ? 1 : code < 22 ? 2 : code < 24 ? 3 : code == 24 ? 4 : code == 25 ? 6 : code - 19;
}
ZHD U32 MLbits(U32 code) // ML_bits[code], code <= 52
{
    return code < 32 ? 0 : code < 36 ? 1 : code < 38 ? 2 : code < 40 ? 3 : code < 42 ? 4 : code == 42 ? 5 : code == 43 ? 7 : code - 36;
}
ZHD U32 LLcode(U32 l) // ZSTD_LLcode
{
    if (l > 63)
        return highbit32(l) + 19;
    return l < 16 ? l : l < 24 ? 16 + ((l - 16) >> 1) : l < 32 ? 20 + ((l - 24) >> 2) : l < 48 ? 22 + ((l - 32) >> 3) : 24;
}
ZHD U32 MLcode(U32 m) // ZSTD_MLcode (m = match length - MINMATCH)
{
    if (m > 127)
        return highbit32(m) + 36;
    return m < 32 ? m : m < 40 ? 32 + ((m - 32) >> 1) : m < 48 ? 36 + ((m - 40) >> 2) : m < 64 ? 38 + ((m - 48) >> 3) : m < 96 ? 40 + ((m - 64) >> 4) : 42;
}

} // namespace zs
