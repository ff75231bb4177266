// scan_kernels.hip -- splitter scan (a2-a4) and FASTA-body preprocessing (a1) for gfx950.
//
// Reference behaviour reproduced (file:line under the reference tree):
//   rolling canonical k-mer   CKmer::insert_canonical / data_canonical   src/core/kmer.h:284-301, 360-362
//   scan loop                 CAGCCompressor::compress_contig            src/core/agc_compressor.cpp:2007-2036
//   membership                bloom_set_t::check && hash_set_lp::check   src/core/utils_adv.h:227-276, src/core/hs.h:489-497
//   symbol codes              preprocess_raw_contig + cnv_num            src/core/agc_compressor.cpp:907-951, src/common/agc_basic.h:40-50
//
// Layout: every lane owns 16 consecutive symbols fetched with ONE coalesced 16-byte load
// (a wave step covers 1 KiB of the contig); the 2-bit packed words and the non-ACGT masks
// of the two preceding lanes arrive by cross-lane shuffles, so each symbol is read from
// HBM exactly once.  The k-mer at the first owned position is assembled from the packed
// words, the other 15 are rolled.  A 64 KiB bloom filter of the splitter set lives in LDS
// (pure accelerator, like the reference's); survivors are checked in an exact
// open-addressing table that stays L2-resident.
//
// The kernel reports EVERY position whose k-mer is a splitter; the reference's "reset the
// k-mer after a hit" rule (a hit voids the next k-1 positions) is a sequential filter over
// the sparse hit list and is applied by the host right after (api.hip: accept_hits).
#include "dev_common.h"

namespace agc {

__device__ __forceinline__ uint32_t pack4(uint32_t w)
{
    // 4 symbols (low 2 bits of each byte, first symbol in the low byte) -> 8 bits, first symbol most significant
    return ((w & 0x03030303u) * 0x40100401u) >> 24;
}

__device__ __forceinline__ uint32_t inv4(uint32_t w)
{
    // bit j set iff byte j is > 3; returned with the FIRST symbol in the most significant of 4 bits
    uint32_t v = w & 0xFCFCFCFCu;
    uint32_t nz = ((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v; // bit 7 of each byte = byte != 0
    nz &= 0x80808080u;
    // gather bits 7,15,23,31 -> 3,2,1,0
    return ((nz >> 7) & 1u) << 3 | ((nz >> 15) & 1u) << 2 | ((nz >> 23) & 1u) << 1 | (nz >> 31);
}

// reverse the order of the 2-bit groups of a 64-bit word
__device__ __forceinline__ uint64_t rev2(uint64_t x)
{
    x = __brevll(x);
    return ((x & 0x5555555555555555ULL) << 1) | ((x >> 1) & 0x5555555555555555ULL);
}

struct ScanArgs {
    const uint8_t *codes;
    const ScanRange *ranges;
    uint32_t n_ranges;
    uint32_t k;
    const uint64_t *table;     // exact splitter table, ~0 = empty
    uint64_t table_mask;
    const uint32_t *bloom;     // BLOOM_WORDS words (copied to LDS)
    const uint32_t *bloom2;    // BLOOM2_WORDS words (global / L2)
    ScanHit *hits;
    uint32_t *n_hits;
    uint32_t cap;
};

__global__ void __launch_bounds__(1024, 8) scan_kernel(ScanArgs a)
{
    __shared__ uint32_t s_bloom[BLOOM_WORDS];
    for (uint32_t t = threadIdx.x; t < BLOOM_WORDS; t += blockDim.x)
        s_bloom[t] = a.bloom[t];
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t waves_per_block = blockDim.x >> 6;
    const uint32_t k = a.k;
    const uint64_t kmask = k == 32 ? ~0ULL : ((1ULL << (2 * k)) - 1ULL);
    const uint32_t lshift = 64 - 2 * k;

    for (uint32_t r = blockIdx.x * waves_per_block + wave; r < a.n_ranges; r += gridDim.x * waves_per_block) {
        const ScanRange rg = a.ranges[r];
        // previous two 16-symbol chunks (packed word + invalid mask) of lanes 62/63 of the
        // previous step; at the start: the 32 symbols before rg.begin (invalid outside the contig)
        uint32_t carryP1 = 0, carryP2 = 0, carryI1 = 0xFFFF, carryI2 = 0xFFFF; // 1 = immediately before
        {
            // lanes 0 and 1 assemble the two halo chunks
            uint32_t P = 0, I = 0xFFFF;
            if (lane < 2) {
                const uint64_t cb = rg.begin - 16 * (uint64_t)(lane + 1); // may wrap below 0: handled per byte
                P = 0;
                I = 0;
#pragma unroll 1
                for (uint32_t j = 0; j < 16; ++j) {
                    const uint64_t p = cb + j;
                    uint32_t c = 4;
                    if (rg.begin >= 16 * (uint64_t)(lane + 1) - j && p >= rg.ctg_begin && p < rg.begin)
                        c = a.codes[p];
                    P = (P << 2) | (c & 3);
                    I = (I << 1) | (c > 3);
                }
            }
            carryP1 = __shfl(P, 0);
            carryI1 = __shfl(I, 0);
            carryP2 = __shfl(P, 1);
            carryI2 = __shfl(I, 1);
        }

        // the 16-byte chunk of the NEXT step is requested before the current one is processed
        uint4 nxt = make_uint4(0, 0, 0, 0);
        {
            const uint64_t off0 = rg.begin + (uint64_t)lane * 16;
            if (off0 + 16 <= rg.end)
                __builtin_memcpy(&nxt, a.codes + off0, 16);
        }
        for (uint64_t base = rg.begin; base < rg.end; base += 1024) {
            const uint64_t off = base + (uint64_t)lane * 16;
            const uint4 v = nxt;
            {
                const uint64_t offn = off + 1024;
                if (offn + 16 <= rg.end)
                    __builtin_memcpy(&nxt, a.codes + offn, 16);
            }
            uint32_t P = 0, I = 0xFFFF;
            uint32_t nvalid = 0; // symbols of this chunk inside the range
            if (off < rg.end) {
                const uint64_t rem = rg.end - off;
                if (rem >= 16) {
                    P = (pack4(v.x) << 24) | (pack4(v.y) << 16) | (pack4(v.z) << 8) | pack4(v.w);
                    I = (inv4(v.x) << 12) | (inv4(v.y) << 8) | (inv4(v.z) << 4) | inv4(v.w);
                    nvalid = 16;
                } else {
                    P = 0;
                    I = 0;
#pragma unroll 1
                    for (uint32_t j = 0; j < 16; ++j) {
                        uint32_t c = 4;
                        if (j < rem)
                            c = a.codes[off + j];
                        P = (P << 2) | (c & 3);
                        I = (I << 1) | (c > 3);
                    }
                    nvalid = (uint32_t)rem;
                }
            }
            // neighbours' chunks
            uint32_t P1 = __shfl_up(P, 1), I1 = __shfl_up(I, 1);
            uint32_t P2 = __shfl_up(P, 2), I2 = __shfl_up(I, 2);
            if (lane == 0) {
                P1 = carryP1;
                I1 = carryI1;
                P2 = carryP2;
                I2 = carryI2;
            } else if (lane == 1) {
                P2 = carryP1;
                I2 = carryI1;
            }
            carryP1 = __shfl(P, 63);
            carryI1 = __shfl(I, 63);
            carryP2 = __shfl(P, 62);
            carryI2 = __shfl(I, 62);

            if (nvalid) {
                // 96-bit window: symbols -32..-17 (P2), -16..-1 (P1), 0..15 (P); symbol j of the own
                // chunk sits at bits [2*(15-j), 2*(15-j)+1] of P.
                const uint64_t hi = ((uint64_t)P2 << 32) | P1;   // symbols -32..-1
                // invalid bits: bit (47 - q) <-> symbol q-32, q in 0..47
                const uint64_t inv = ((uint64_t)I2 << 32) | ((uint64_t)I1 << 16) | I;
                const uint64_t wmask = k == 32 ? 0xFFFFFFFFULL : ((1ULL << k) - 1ULL);

                // k-mer ending at own symbol 0: the k symbols -(k-1)..0, right aligned
                const uint64_t dir0 = ((hi << 2) | (P >> 30)) & kmask;
                // reverse complement: ~dir flips every symbol to 3-s, rev2 puts the last symbol first
                const uint64_t rc0 = (rev2(~dir0) >> lshift) & kmask;

                // pass 1 (branch-free, unrolled): bloom test of the 16 k-mers.  Both forms are kept
                // LEFT-aligned in 32-bit halves (as CKmer does, kmer.h:284-301): rolling is then two
                // funnel shifts, nothing shifts out of the field and no 64-bit VALU op is needed.
                uint32_t pass = 0;
                {
                    const uint64_t dl0 = dir0 << lshift, rl0 = rc0 << lshift;
                    uint32_t dh = (uint32_t)(dl0 >> 32), dl = (uint32_t)dl0, rh = (uint32_t)(rl0 >> 32), rl = (uint32_t)rl0;
                    const uint32_t lo_mask = lshift ? (~0u << lshift) : ~0u; // lshift = 64-2k in [0,30]
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        uint32_t bw[8], bm[8];
#pragma unroll
                        for (int t = 0; t < 8; ++t) {
                            const int j = g * 8 + t;
                            if (j) {
                                const uint32_t sym = (P >> (2 * (15 - j))) & 3u;
                                dh = __builtin_amdgcn_alignbit(dh, dl, 30);          // (dh << 2) | (dl >> 30)
                                dl = (dl << 2) | (sym << lshift);
                                rl = __builtin_amdgcn_alignbit(rh, rl, 2) & lo_mask; // (rl >> 2) | (rh << 30)
                                rh = (rh >> 2) | ((sym ^ 3u) << 30);
                            }
                            const bool d_lt = dh < rh || (dh == rh && dl < rl);
                            bloom_slot(d_lt ? dh : rh, d_lt ? dl : rl, bw[t], bm[t]);
                        }
#pragma unroll
                        for (int t = 0; t < 8; ++t)
                            pass |= (uint32_t)((s_bloom[bw[t]] & bm[t]) == bm[t]) << (g * 8 + t);
                    }
                }
                // window validity: symbols (j-k+1 .. j) <-> inv bits (15-j) .. (15-j+k-1); j < nvalid
                uint32_t ok = 0xFFFFu;
                if (inv) {
                    ok = 0;
#pragma unroll 1
                    for (int j = 0; j < 16; ++j)
                        ok |= (uint32_t)(((inv >> (15 - j)) & wmask) == 0) << j;
                }
                pass &= ok & ((1u << nvalid) - 1u);

                // pass 2: second-level filter + exact table for the bloom survivors.  Rare per lane but some
                // lane of the wave almost always has one, so it must be cheap: each survivor's k-mer is cut
                // straight out of the packed window (no re-rolling), and the loop runs over set bits only.
                if (pass) {
                    const uint64_t w_lo = (hi << 32) | P;   // symbols -16..15
                    const uint64_t w_hi = hi >> 32;          // symbols -32..-17
                    while (pass) {
                        const uint32_t j = (uint32_t)__builtin_ctz(pass);
                        pass &= pass - 1;
                        const uint32_t sft = 2 * (15 - j);
                        uint64_t dir = w_lo >> sft;
                        if (sft)
                            dir |= w_hi << (64 - sft);
                        dir &= kmask;
                        const uint64_t rcv = (rev2(~dir) >> lshift) & kmask;
                        const uint64_t dl = dir << lshift, rl = rcv << lshift;
                        const uint64_t can = dl < rl ? dl : rl;
                        const uint64_t h = splitter_hash(can);
                        uint32_t w2, m2;
                        bloom2_slot(h, w2, m2);
                        if ((a.bloom2[w2] & m2) != m2)
                            continue;
                        uint64_t slot = h & a.table_mask;
                        for (;;) {
                            const uint64_t e = a.table[slot];
                            if (e == can) {
                                const uint32_t idx = atomicAdd(a.n_hits, 1u);
                                if (idx < a.cap) {
                                    a.hits[idx].pos = off + j;
                                    a.hits[idx].dir = dl;
                                    a.hits[idx].rc = rl;
                                }
                                break;
                            }
                            if (e == ~0ULL)
                                break;
                            slot = (slot + 1) & a.table_mask;
                        }
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// 2-bit packed samples: pack / expand / scan
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pack4le(uint32_t w)
{
    // 4 symbols (low 2 bits of each byte, first symbol in the low byte) -> 8 bits, first symbol in the LOW bits
    uint32_t x = w & 0x03030303u;
    x = (x | (x >> 6)) & 0x000F000Fu;
    return (x | (x >> 12)) & 0xFFu;
}
__device__ __forceinline__ uint32_t unpack4le(uint32_t b)
{
    uint32_t x = b & 0xFFu;
    x = (x | (x << 12)) & 0x000F000Fu;
    return (x | (x << 6)) & 0x03030303u;
}
// reverse the order of the sixteen 2-bit groups of a word
__device__ __forceinline__ uint32_t rev2_32(uint32_t x)
{
    x = __brev(x);
    return ((x & 0x55555555u) << 1) | ((x >> 1) & 0x55555555u);
}

// codes (1 B per symbol) -> packed words + escaped blocks; one wave per block of PACK_BLOCK symbols
__global__ void __launch_bounds__(256) pack_codes_kernel(const uint8_t *__restrict__ codes, uint64_t n, uint32_t *__restrict__ words,
                                                         int32_t *__restrict__ esc_index, uint8_t *__restrict__ esc_bytes,
                                                         uint32_t *__restrict__ esc_count, uint32_t esc_cap)
{
    const uint64_t n_blocks = (n + PACK_BLOCK - 1) / PACK_BLOCK;
    const uint32_t lane = threadIdx.x & 63;
    for (uint64_t blk = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); blk < n_blocks; blk += (uint64_t)gridDim.x * 4) {
        const uint64_t g = blk * PACK_BLOCK + (uint64_t)lane * 16;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (g + 16 <= n)
            __builtin_memcpy(&v, codes + g, 16);
        else if (g < n) {
            uint8_t tmp[16];
            for (uint32_t j = 0; j < 16; ++j)
                tmp[j] = g + j < n ? codes[g + j] : 0;
            __builtin_memcpy(&v, tmp, 16);
        }
        const bool high = ((v.x | v.y | v.z | v.w) & 0xFCFCFCFCu) != 0;
        const uint64_t any = __ballot(high);
        int32_t slot = -1;
        if (any) {
            uint32_t s = 0;
            if (lane == 0)
                s = atomicAdd(esc_count, 1u);
            s = (uint32_t)__shfl((int)s, 0);
            if (s < esc_cap) {
                slot = (int32_t)s;
                *(uint4 *)(esc_bytes + (uint64_t)s * PACK_BLOCK + lane * 16) = v;
            } else
                slot = -2; // over capacity: the caller retries with a larger buffer
        }
        if (lane == 0)
            esc_index[blk] = slot;
        // (escaped blocks: symbols outside ACGT leave their two low bits, which nobody reads)
        words[g >> 4] = pack4le(v.x) | (pack4le(v.y) << 8) | (pack4le(v.z) << 16) | (pack4le(v.w) << 24);
    }
}

// packed -> codes (1 B per symbol), the staging form the LZ kernels read; one wave per block
__global__ void __launch_bounds__(256) expand_codes_kernel(PackedView pv, uint8_t *__restrict__ codes)
{
    const uint64_t n = pv.n_symbols;
    const uint64_t n_blocks = (n + PACK_BLOCK - 1) / PACK_BLOCK;
    const uint32_t lane = threadIdx.x & 63;
    for (uint64_t blk = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6); blk < n_blocks; blk += (uint64_t)gridDim.x * 4) {
        const uint64_t g = blk * PACK_BLOCK + (uint64_t)lane * 16;
        if (g >= n)
            continue;
        const int32_t slot = pv.esc_index[blk];
        uint4 v;
        if (slot >= 0)
            v = *(const uint4 *)(pv.esc_bytes + (uint64_t)slot * PACK_BLOCK + lane * 16);
        else {
            const uint32_t w = pv.words[g >> 4];
            v = make_uint4(unpack4le(w), unpack4le(w >> 8), unpack4le(w >> 16), unpack4le(w >> 24));
        }
        if (g + 16 <= n)
            *(uint4 *)(codes + g) = v; // (the output buffer is 16-byte aligned)
        else {
            uint8_t tmp[16];
            __builtin_memcpy(tmp, &v, 16);
            for (uint32_t j = 0; g + j < n; ++j)
                codes[g + j] = tmp[j];
        }
    }
}

// ---- splitter scan over a packed sample ----
// Same contract as scan_kernel (every position whose canonical k-mer is a splitter), different first-level test: a k-mer
// can only be a splitter if its LAST 16 SYMBOLS are the last 16 symbols of a splitter or of a splitter's reverse
// complement, and that 32-bit word costs one funnel shift of the packed stream per position -- no rolling of two strands,
// no canonical select.  A 128 KiB filter over those words lives in LDS (one 1024-thread block per CU); the survivors
// (~3 %) get the full k-mer, its canonical form, the second-level filter and the exact table as before.
typedef __attribute__((address_space(3))) const uint32_t lds_cu32;
struct ScanPackedArgs {
    PackedView pv;
    const ScanRange *ranges;   // begin / end multiples of 16 symbols relative to the buffer, except at contig ends
    uint32_t n_ranges;
    uint32_t k;
    const uint64_t *table;
    uint64_t table_mask;
    const uint32_t *sbloom;    // SBLOOM_WORDS words (copied to LDS)
    const uint32_t *bloom2;
    ScanHit *hits;
    uint32_t *n_hits;
    uint32_t cap;
};

// 16 symbols at the 16-aligned buffer position g (may lie before 0): P = 2-bit codes, first symbol most significant;
// I = invalid mask (first symbol = bit 15): outside ACGT or outside the contig [cb, ce)
__device__ __forceinline__ void load_chunk(const PackedView &pv, int64_t g, uint64_t cb, uint64_t ce, uint32_t &P, uint32_t &I)
{
    P = 0;
    I = 0xFFFF;
    if (g + 16 <= (int64_t)cb || g >= (int64_t)ce)
        return;
    const int32_t slot = pv.esc_index[(uint64_t)g / PACK_BLOCK];
    if (slot < 0) {
        P = rev2_32(pv.words[(uint64_t)g >> 4]);
        I = 0;
    } else {
        const uint4 v = *(const uint4 *)(pv.esc_bytes + (uint64_t)slot * PACK_BLOCK + ((uint64_t)g & (PACK_BLOCK - 1)));
        P = (pack4(v.x) << 24) | (pack4(v.y) << 16) | (pack4(v.z) << 8) | pack4(v.w);
        I = (inv4(v.x) << 12) | (inv4(v.y) << 8) | (inv4(v.z) << 4) | inv4(v.w);
    }
    if ((int64_t)cb > g) // the first cb - g symbols belong to the previous contig
        I |= ~(0xFFFFu >> (uint32_t)((int64_t)cb - g)) & 0xFFFFu;
    if ((int64_t)ce < g + 16) // symbols from ce on belong to the next one
        I |= 0xFFFFu >> (uint32_t)((int64_t)ce - g);
}

__global__ void __launch_bounds__(1024, 4) scan_packed_kernel(ScanPackedArgs a)
{
    extern __shared__ uint32_t s_sbloom[];
    for (uint32_t t = threadIdx.x; t < SBLOOM_WORDS; t += blockDim.x)
        s_sbloom[t] = a.sbloom[t];
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t waves_per_block = blockDim.x >> 6;
    const uint32_t k = a.k;
    const uint64_t kmask = k == 32 ? ~0ULL : ((1ULL << (2 * k)) - 1ULL);
    const uint32_t lshift = 64 - 2 * k;
    const uint64_t wmask = k == 32 ? 0xFFFFFFFFULL : ((1ULL << k) - 1ULL);

    for (uint32_t r_ = blockIdx.x * waves_per_block + wave; r_ < a.n_ranges; r_ += gridDim.x * waves_per_block) {
        // (a range per wave: its descriptor, the step's base and everything compared with them live in scalar registers)
        const uint32_t r = __builtin_amdgcn_readfirstlane(r_);
        const ScanRange rg = a.ranges[r];
        const int64_t first = (int64_t)(rg.begin & ~(uint64_t)(PACK_BLOCK - 1));
        uint32_t carryP1, carryP2, carryI1, carryI2; // chunks right before the step (1 = immediately before)
        {
            uint32_t P = 0, I = 0xFFFF;
            if (lane < 2)
                load_chunk(a.pv, first - 16 * (int64_t)(lane + 1), rg.ctg_begin, rg.ctg_end, P, I);
            carryP1 = __builtin_amdgcn_readlane(P, 0);
            carryI1 = __builtin_amdgcn_readlane(I, 0);
            carryP2 = __builtin_amdgcn_readlane(P, 1);
            carryI2 = __builtin_amdgcn_readlane(I, 1);
        }
        for (int64_t base = first; base < (int64_t)rg.end; base += PACK_BLOCK) {
            const int64_t g = base + (int64_t)lane * 16;
            uint32_t P, I;
            // (a step whose 1024 symbols lie inside the contig, in a block without escapes -- all but a contig's ends and its
            // N runs --: the lane's word, nothing to mask)
            const bool plain = base >= (int64_t)rg.ctg_begin && base + (int64_t)PACK_BLOCK <= (int64_t)rg.ctg_end &&
                               a.pv.esc_index[(uint64_t)base / PACK_BLOCK] < 0;
            if (plain) {
                P = rev2_32(a.pv.words[(uint64_t)g >> 4]);
                I = 0;
            } else
                load_chunk(a.pv, g, rg.ctg_begin, rg.ctg_end, P, I);
            // the two chunks before the lane's own: the lane below's, by a wave-wide DPP shift (lane 0 keeps the carry it is
            // given as the old value: no shuffle through the LDS, no select)
            const uint32_t P1 = __builtin_amdgcn_update_dpp(carryP1, P, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
            const uint32_t I1 = __builtin_amdgcn_update_dpp(carryI1, I, 0x138, 0xF, 0xF, false);
            const uint32_t P2 = __builtin_amdgcn_update_dpp(carryP2, P1, 0x138, 0xF, 0xF, false);
            const uint32_t I2 = __builtin_amdgcn_update_dpp(carryI2, I1, 0x138, 0xF, 0xF, false);
            carryP1 = __builtin_amdgcn_readlane(P, 63);
            carryI1 = __builtin_amdgcn_readlane(I, 63);
            carryP2 = __builtin_amdgcn_readlane(P, 62);
            carryI2 = __builtin_amdgcn_readlane(I, 62);

            // positions of this chunk the range reports: begin <= g + j < end
            uint32_t vmask = 0xFFFFu;
            if (!(base >= (int64_t)rg.begin && base + (int64_t)PACK_BLOCK <= (int64_t)rg.end)) { // (a range's first and last step)
                if ((int64_t)rg.begin > g)
                    vmask &= (int64_t)rg.begin - g >= 16 ? 0u : (0xFFFFu << (uint32_t)((int64_t)rg.begin - g));
                if ((int64_t)rg.end < g + 16)
                    vmask &= (int64_t)rg.end <= g ? 0u : (0xFFFFu >> (uint32_t)(g + 16 - (int64_t)rg.end));
            }
            // (no lane leaves the step early: the survivors below are dealt out over ALL lanes of the wave)

            // pass 1: filter on the last 16 symbols of the k-mer ending at every own position
            // (position 15 first: `pass` is doubled and takes the test's outcome as the carry -- v_cmp + v_addc, two
            // instructions where a select, a shift and an or were three; the filter word is read through an LDS pointer made
            // from the byte address itself: the dynamic LDS of this kernel starts at 0, and the compiler otherwise adds the base)
            uint32_t pass = 0;
#pragma unroll
            for (int h = 1; h >= 0; --h) { // (eight reads in flight before the first outcome is needed)
                uint32_t fw[8], bm[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int j = 8 * h + i;
                    const uint32_t w16 = j == 15 ? P : __builtin_amdgcn_alignbit(P1, P, 2 * (15 - j)); // symbols j-15 .. j
                    uint32_t ba;
                    sbloom_addr_mask(w16, ba, bm[i]);
                    fw[i] = *(lds_cu32 *)(uintptr_t)ba;
                }
#pragma unroll
                for (int i = 7; i >= 0; --i) {
                    const uint32_t missing = bm[i] & ~fw[i];
                    asm("v_cmp_eq_u32_e32 vcc, 0, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(pass) : "v"(missing) : "vcc");
                }
            }
            pass &= vmask;
            // pass 2: the survivors (2 % of the positions: 20 per step of a wave, 0-3 per lane) get the full k-mer, the second
            // filter and the exact table.  Each lane used to walk its own: as many rounds as the unluckiest lane has survivors
            // (2-3), each with a fraction of the lanes at work and a dependent load from the second filter -- that, not the
            // filter pass, was most of the kernel's time.  Now the survivors of the wave are numbered (prefix sum of the lanes'
            // counts) and lane t takes survivor t: it finds the owner by a binary search over the prefix sums, the owner's
            // windows by shuffles -- one round for up to 64 survivors.
            if (__ballot(pass != 0)) {
                const uint32_t cnt = (uint32_t)__popc(pass);
                uint32_t incl = cnt; // inclusive prefix sum over the wave: four steps inside the rows of 16, two across them
                incl += __builtin_amdgcn_update_dpp(0u, incl, 0x111 /* row_shr:1 */, 0xF, 0xF, false);
                incl += __builtin_amdgcn_update_dpp(0u, incl, 0x112 /* row_shr:2 */, 0xF, 0xF, false);
                incl += __builtin_amdgcn_update_dpp(0u, incl, 0x114 /* row_shr:4 */, 0xF, 0xF, false);
                incl += __builtin_amdgcn_update_dpp(0u, incl, 0x118 /* row_shr:8 */, 0xF, 0xF, false);
                incl += __builtin_amdgcn_update_dpp(0u, incl, 0x142 /* row_bcast:15 */, 0xA, 0xF, false);
                incl += __builtin_amdgcn_update_dpp(0u, incl, 0x143 /* row_bcast:31 */, 0xC, 0xF, false);
                const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
                for (uint32_t tb = 0; tb < total; tb += 64) {
                    const uint32_t t = tb + lane;
                    uint32_t src = 0; // the first lane whose inclusive count exceeds t
#pragma unroll
                    for (int b = 32; b; b >>= 1) {
                        const uint32_t v = __shfl(incl, (int)(src + b - 1));
                        if (v <= t)
                            src += b;
                    }
                    if (src > 63)
                        src = 63; // (t >= total: not a survivor, masked below)
                    uint32_t r = t - (__shfl(incl, (int)src) - __shfl(cnt, (int)src)); // its rank among the owner's survivors
                    uint32_t m = __shfl(pass, (int)src);
                    const uint32_t oP = __shfl(P, (int)src), oP1 = __shfl(P1, (int)src), oP2 = __shfl(P2, (int)src);
                    const uint32_t oI = __shfl(I, (int)src), oI1 = __shfl(I1, (int)src), oI2 = __shfl(I2, (int)src);
                    if (t >= total)
                        continue;
                    for (; r; --r)
                        m &= m - 1;
                    const uint32_t j = (uint32_t)__builtin_ctz(m);
                    // window validity: symbols (j-k+1 .. j) <-> inv bits (15-j) .. (15-j+k-1)
                    const uint64_t inv = ((uint64_t)oI2 << 32) | ((uint64_t)oI1 << 16) | oI;
                    if (((inv >> (15 - j)) & wmask) != 0)
                        continue;
                    const uint64_t hi = ((uint64_t)oP2 << 32) | oP1; // symbols -32..-1
                    const uint64_t w_lo = (hi << 32) | oP;            // symbols -16..15
                    const uint64_t w_hi = hi >> 32;                   // symbols -32..-17
                    const uint32_t sft = 2 * (15 - j);
                    uint64_t dir = w_lo >> sft;
                    if (sft)
                        dir |= w_hi << (64 - sft);
                    dir &= kmask;
                    const uint64_t rcv = (rev2(~dir) >> lshift) & kmask;
                    const uint64_t dl = dir << lshift, rl = rcv << lshift;
                    const uint64_t can = dl < rl ? dl : rl;
                    const uint64_t h = splitter_hash(can);
                    uint32_t w2, m2;
                    bloom2_slot(h, w2, m2);
                    if ((a.bloom2[w2] & m2) != m2)
                        continue;
                    uint64_t slot = h & a.table_mask;
                    for (;;) {
                        const uint64_t e = a.table[slot];
                        if (e == can) {
                            const uint32_t idx = atomicAdd(a.n_hits, 1u);
                            if (idx < a.cap) {
                                a.hits[idx].pos = (uint64_t)(base + (int64_t)src * 16) + j;
                                a.hits[idx].dir = dl;
                                a.hits[idx].rc = rl;
                            }
                            break;
                        }
                        if (e == ~0ULL)
                            break;
                        slot = (slot + 1) & a.table_mask;
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// a1: raw FASTA body -> codes.  Three passes: per-block kept counts, scan of the block
// counts (single block), scatter.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint8_t cnv_symbol(uint8_t c)
{
    // cnv_num, src/common/agc_basic.h:40-50 (only consulted for c >= 64; c < 128)
    c &= 127;
    if (c == 64 || c == 96)
        return 32;
    const uint8_t u = c & 0x5F; // fold case: 'a'..'z' -> 'A'..'Z'; 123..127 -> 91..95
    switch (u) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    case 'N': return 4;
    case 'R': return 5;
    case 'Y': return 6;
    case 'S': return 7;
    case 'W': return 8;
    case 'K': return 9;
    case 'M': return 10;
    case 'B': return 11;
    case 'D': return 12;
    case 'H': return 13;
    case 'V': return 14;
    case 'U': return 15;
    default: return 30;
    }
}

constexpr uint32_t PP_TILE = 16384; // bytes per block

__global__ void __launch_bounds__(256) pp_count_kernel(const uint8_t *__restrict__ raw, uint64_t n, uint32_t *__restrict__ block_cnt)
{
    const uint64_t b0 = (uint64_t)blockIdx.x * PP_TILE;
    uint32_t c = 0;
    for (uint32_t t = threadIdx.x * 16; t < PP_TILE; t += blockDim.x * 16) {
        const uint64_t p = b0 + t;
        if (p + 16 <= n) {
            uint4 v;
            __builtin_memcpy(&v, raw + p, 16);
            c += __popc(((v.x >> 6) | (v.x >> 7)) & 0x01010101u);
            c += __popc(((v.y >> 6) | (v.y >> 7)) & 0x01010101u);
            c += __popc(((v.z >> 6) | (v.z >> 7)) & 0x01010101u);
            c += __popc(((v.w >> 6) | (v.w >> 7)) & 0x01010101u);
        } else {
            for (uint32_t j = 0; j < 16 && p + j < n; ++j)
                c += (raw[p + j] >> 6) != 0;
        }
    }
    for (int o = 32; o > 0; o >>= 1)
        c += __shfl_down(c, o);
    __shared__ uint32_t part[4];
    if ((threadIdx.x & 63) == 0)
        part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0)
        block_cnt[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// exclusive scan of block counts into 64-bit offsets; one block
__global__ void __launch_bounds__(1024) pp_scan_kernel(const uint32_t *__restrict__ block_cnt, uint32_t n_blocks,
                                                       uint64_t *__restrict__ block_off, uint64_t *__restrict__ total)
{
    __shared__ uint64_t s_part[1024];
    const uint32_t per = (n_blocks + blockDim.x - 1) / blockDim.x;
    const uint32_t b = threadIdx.x * per, e = min(b + per, n_blocks);
    uint64_t sum = 0;
    for (uint32_t i = b; i < e; ++i)
        sum += block_cnt[i];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t acc = 0;
        for (uint32_t i = 0; i < blockDim.x; ++i) {
            const uint64_t v = s_part[i];
            s_part[i] = acc;
            acc += v;
        }
        *total = acc;
    }
    __syncthreads();
    uint64_t acc = s_part[threadIdx.x];
    for (uint32_t i = b; i < e; ++i) {
        block_off[i] = acc;
        acc += block_cnt[i];
    }
}

__global__ void __launch_bounds__(256) pp_scatter_kernel(const uint8_t *__restrict__ raw, uint64_t n,
                                                         const uint64_t *__restrict__ block_off, uint8_t *__restrict__ codes)
{
    // each thread owns 64 consecutive input bytes of the tile; ranks come from a block scan
    const uint64_t b0 = (uint64_t)blockIdx.x * PP_TILE;
    const uint64_t p0 = b0 + (uint64_t)threadIdx.x * 64;
    uint8_t buf[64];
    uint32_t c = 0;
    for (uint32_t j = 0; j < 64; ++j) {
        const uint64_t p = p0 + j;
        uint8_t x = p < n ? raw[p] : 0;
        buf[j] = x;
        c += (x >> 6) != 0;
    }
    // exclusive scan of c over the block
    __shared__ uint32_t s_w[4];
    uint32_t incl = c;
    const uint32_t lane = threadIdx.x & 63;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o);
        if (lane >= (uint32_t)o)
            incl += v;
    }
    if (lane == 63)
        s_w[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t wbase = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w)
        wbase += s_w[w];
    uint64_t o = block_off[blockIdx.x] + wbase + incl - c;
    for (uint32_t j = 0; j < 64; ++j) {
        const uint8_t x = buf[j];
        if (x >> 6)
            codes[o++] = cnv_symbol(x);
    }
}

} // namespace agc
