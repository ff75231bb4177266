// sym_view.h -- symbol access on the 2-bit packed layout (gfx950 kernels; the same source compiles for the host so that
// tests/symview_host checks every accessor against plain byte arrays on the CPU).
//
// The layout is the sample's own (dev_common.h: PackedView): symbol s of a buffer at bits [2 (s & 15), +1] of 32-bit word
// s >> 4; 1024-symbol blocks holding anything outside ACGT are "escaped" -- kept one byte per symbol in esc_bytes,
// esc_index[block] = slot there or < 0.  The reference keeps sequences one byte per symbol and forms 2-bit codes on the fly
// (get_code, src/common/lz_diff.h:58-106; bytes2tuples, src/common/segment.h:73-138); here the 2-bit form IS what sits in HBM,
// for samples and for group references alike, and the LZ kernels compare 32 symbols per 64-bit XOR.
//
// A SymView names one sequence inside such a buffer, optionally read reverse-complemented
// (reverse_complement_copy, src/common/agc_basic.cpp:282-315: order reversed, c < 4 -> 3 - c, everything else unchanged).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SV_HD __host__ __device__ __forceinline__
#else
#define SV_HD inline
#endif

namespace agc {

constexpr uint32_t SV_BLOCK = 1024; // == PACK_BLOCK

struct SymView {
    const uint32_t *words;    // the buffer's 2-bit words
    const int32_t *esc_index; // per block of the buffer; nullptr: the buffer has no escaped block at all
    const uint8_t *esc_bytes;
    uint64_t start;           // buffer index of the sequence's first stored symbol
    uint32_t len;
    uint32_t rc;              // read as the reverse complement: position p = comp(buffer[start + len - 1 - p])
};

SV_HD uint32_t sv_brev32(uint32_t x)
{
#if defined(__clang__)
    return __builtin_bitreverse32(x);
#else
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    return __builtin_bswap32(x);
#endif
}

SV_HD uint64_t sv_brev64(uint64_t x)
{
#if defined(__clang__)
    return __builtin_bitreverse64(x);
#else
    return ((uint64_t)sv_brev32((uint32_t)x) << 32) | sv_brev32((uint32_t)(x >> 32));
#endif
}

// order of the 2-bit fields reversed (field 0 <-> field 31)
SV_HD uint64_t sv_rev2_64(uint64_t x)
{
    x = sv_brev64(x);
    return ((x >> 1) & 0x5555555555555555ULL) | ((x & 0x5555555555555555ULL) << 1);
}

SV_HD uint32_t sv_ctz64(uint64_t x) { return (uint32_t)__builtin_ctzll(x); }

// (The accessors are templates over the view type V -- any struct with SymView's members -- and over the word pointer type: a kernel
// that has loaded a view out of a descriptor holds GENERIC pointers as far as the compiler can tell and would get FLAT loads for
// every access; it converts the view once into one whose pointers name the global address space, lz_kernels.hip: SymViewG.)
// 32 stored symbols starting at buffer index s (symbol s + j at bits [2j, 2j+1]); reads words s/16 .. s/16 + 2
template <class W> SV_HD uint64_t sv_raw64(W words, uint64_t s)
{
    const W p = words + (s >> 4);
    const uint32_t sh = 2u * (uint32_t)(s & 15u);
    const uint32_t a = p[0], b = p[1], c = p[2];
    uint64_t x = (((uint64_t)b << 32) | a) >> sh;
    if (sh)
        x |= (uint64_t)c << (64 - sh);
    return x;
}

// one symbol of the buffer (full code, escapes looked up)
template <class V> SV_HD uint32_t sv_buf_sym(const V &v, uint64_t s, bool clean)
{
    if (!clean && v.esc_index) {
        const int32_t slot = v.esc_index[s / SV_BLOCK];
        if (slot >= 0)
            return v.esc_bytes[(uint64_t)slot * SV_BLOCK + (s & (SV_BLOCK - 1))];
    }
    return (v.words[s >> 4] >> (2u * (uint32_t)(s & 15u))) & 3u;
}

// symbol at position p of the sequence (p < len); clean = the caller knows that no block of the sequence is escaped
template <class V> SV_HD uint32_t sv_sym(const V &v, uint32_t p, bool clean = false)
{
    if (v.rc) {
        const uint32_t c = sv_buf_sym(v, v.start + (v.len - 1u - p), clean);
        return c < 4 ? 3u - c : c;
    }
    return sv_buf_sym(v, v.start + p, clean);
}

// cnt (1..32) symbols from position p (p + cnt <= len): P = their 2-bit codes (position p + j at bits [2j, 2j+1]; symbols
// outside ACGT leave two arbitrary bits), I = bit j set where the symbol is outside ACGT.  Bits of positions >= cnt are
// unspecified in P and clear in I.
template <class V> SV_HD void sv_fetch32(const V &v, uint32_t p, uint32_t cnt, bool clean, uint64_t &P, uint32_t &I)
{
    I = 0;
    uint64_t s0, s1; // first / last buffer index touched
    if (v.rc) {
        s1 = v.start + (v.len - 1u - p);
        s0 = s1 - (cnt - 1u);
    } else {
        s0 = v.start + p;
        s1 = s0 + (cnt - 1u);
    }
    bool fast = clean || !v.esc_index;
    if (!fast) {
        const int32_t e0 = v.esc_index[s0 / SV_BLOCK], e1 = v.esc_index[s1 / SV_BLOCK];
        fast = e0 < 0 && e1 < 0;
    }
    if (fast) {
        if (v.rc) {
            // buffer symbols s1 - 31 .. s1, reversed and complemented (3 - c == c ^ 3)
            const uint64_t x = s1 >= 31 ? sv_raw64(v.words, s1 - 31) : sv_raw64(v.words, 0) << (2u * (31u - (uint32_t)s1));
            P = ~sv_rev2_64(x);
        } else
            P = sv_raw64(v.words, s0);
        return;
    }
    P = 0;
    for (uint32_t j = 0; j < cnt; ++j) {
        const uint32_t c = sv_sym(v, p + j, false);
        P |= (uint64_t)(c & 3u) << (2u * j);
        I |= (uint32_t)(c > 3u) << j;
    }
}

// the same for cnt (1..16) symbols: two dwords per fetch instead of three (P in the low 32 bits)
template <class W> SV_HD uint32_t sv_raw32(W words, uint64_t s)
{
    const W p = words + (s >> 4);
    const uint32_t sh = 2u * (uint32_t)(s & 15u);
    return (uint32_t)(((((uint64_t)p[1]) << 32) | p[0]) >> sh);
}
SV_HD uint32_t sv_rev2_32(uint32_t x)
{
    x = sv_brev32(x);
    return ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
}
template <class V> SV_HD void sv_fetch16(const V &v, uint32_t p, uint32_t cnt, bool clean, uint64_t &P, uint32_t &I)
{
    I = 0;
    uint64_t s0, s1;
    if (v.rc) {
        s1 = v.start + (v.len - 1u - p);
        s0 = s1 - (cnt - 1u);
    } else {
        s0 = v.start + p;
        s1 = s0 + (cnt - 1u);
    }
    bool fast = clean || !v.esc_index;
    if (!fast) {
        const int32_t e0 = v.esc_index[s0 / SV_BLOCK], e1 = v.esc_index[s1 / SV_BLOCK];
        fast = e0 < 0 && e1 < 0;
    }
    if (fast) {
        if (v.rc) {
            const uint32_t x = s1 >= 15 ? sv_raw32(v.words, s1 - 15) : sv_raw32(v.words, 0) << (2u * (15u - (uint32_t)s1));
            P = (uint32_t)~sv_rev2_32(x);
        } else
            P = sv_raw32(v.words, s0);
        return;
    }
    P = 0;
    for (uint32_t j = 0; j < cnt; ++j) {
        const uint32_t c = sv_sym(v, p + j, false);
        P |= (uint64_t)(c & 3u) << (2u * j);
        I |= (uint32_t)(c > 3u) << j;
    }
}
template <uint32_t N, class V> SV_HD void sv_fetch(const V &v, uint32_t p, uint32_t cnt, bool clean, uint64_t &P, uint32_t &I)
{
    if (N == 16)
        sv_fetch16(v, p, cnt, clean, P, I);
    else
        sv_fetch32(v, p, cnt, clean, P, I);
}

// four packed symbols (8 bits) -> four bytes
SV_HD uint32_t sv_expand4(uint32_t x)
{
    x &= 0xFFu;
    return (x | (x << 6) | (x << 12) | (x << 18)) & 0x03030303u;
}

// key of key_len symbols (<= 29) whose packed form is P (position 0 in the low bits): first symbol most significant, as
// get_code builds it (src/common/lz_diff.h:58-106)
SV_HD uint64_t sv_key_from_packed(uint64_t P, uint32_t key_len) { return sv_rev2_64(P) >> (64u - 2u * key_len); }

} // namespace agc
