// pack_kernels.hip -- raw FASTA bodies in HBM -> the 2-bit packed sample without an intermediate one-byte-per-symbol form.
//
// What it replaces: CAGCCompressor::preprocess_raw_contig (src/core/agc_compressor.cpp:907-951: every byte < 64 is dropped -- line
// ends --, the others go through cnv_num, src/common/agc_basic.h:39-49) followed by the packing of the codes (pack_codes_kernel):
// two kernels that read and write one byte per symbol three times over (count, scatter, pack).  Here the raw bytes (1 + 1 / line_width
// per symbol) are read twice and 0.25 bytes per symbol are written.
//
// Shape: a compaction needs every tile's output offset = the number of symbols kept in front of it: a counting pass over the raw
// bytes first (pack_fasta_count_kernel: one count per tile), a two-level scan of the counts (pack_fasta_scan_kernel + pp_scan_kernel
// over the scan blocks' totals), then the pack pass -- the input is read twice, 2 (1 + 1 / width) + 0.25 bytes per symbol of traffic.
// (A ONE-pass variant -- tiles in ticket order publishing their counts through a chained scan with decoupled look-back, relaxed
// device-scope atomics -- was built first and measured slower on MI355X: 3.1 ms per 3 Gbp against 1.9, 0.5 ms of it the 186 k
// same-address ticket atomics, the rest the chain of round trips through eight L2s; with an acquire / release pair per tile, which
// writes back and invalidates the XCD's L2 each time, 19.7 ms.  It lived behind AGC_HIP_PACK_LOOKBACK until commit 9ace5d7.)
// A tile of 16 KiB of input then writes WHOLE 1024-symbol blocks only: it owns the blocks whose first
// symbol it holds, leaves the symbols in front of its first block to the tile before it and reads on into the next tiles'
// bytes (<= 1023 symbols, L2 hits for the neighbour) to finish its last block -- words, escape index and escaped bytes of a block
// have one writer, nothing is zeroed beforehand and no atomics touch the output.
#pragma once

namespace agc {

constexpr uint32_t PF_TILE = 16384;                    // input bytes per tile: 256 threads x 4 chunks x 16 B
constexpr uint32_t PF_MAX_BLOCKS = PF_TILE / PACK_BLOCK + 1; // blocks a tile can own (16; + 1 of slack)

struct PackFastaArgs {
    const uint8_t *raw;
    uint64_t n_raw;
    const uint64_t *rng_begin, *rng_end; // contig c = raw bytes [rng_begin[c], rng_end[c]); ascending, disjoint
    uint32_t n_rng;
    uint32_t n_tiles;
    const uint32_t *tile_local;          // two-pass variant: symbols in front of a tile inside its scan block (pack_fasta_scan_kernel) ...
    const uint64_t *block_off;           // ... and in front of that scan block (pp_scan_kernel over the blocks' totals)
    uint32_t *words;
    int32_t *esc_index;
    uint8_t *esc_bytes;
    uint32_t *esc_count;                 // zeroed
    uint32_t esc_cap;
    unsigned long long *ctg_off;         // n_rng + 1 symbol offsets (all-ones where no tile holds the contig's first byte: host)
    unsigned long long *total;           // symbols in all
};

// one chunk of 16 raw bytes: which of them are symbols, their 2-bit codes squeezed together, whether any lies outside ACGT
struct PfChunk {
    uint32_t bits;  // 2 bits per KEPT symbol, first symbol lowest
    uint32_t cnt;   // kept symbols (<= 16)
    uint32_t keep;  // bit j: raw byte j is a symbol
    uint32_t other; // bit r: the r-th KEPT symbol lies outside ACGT (0: none)
};

// byte j of the four words (a select chain: a dynamic index would send the words to scratch memory)
__device__ __forceinline__ uint8_t pf_byte(const uint32_t w[4], uint32_t j)
{
    const uint32_t q = j >> 2, x = q == 0 ? w[0] : q == 1 ? w[1] : q == 2 ? w[2] : w[3];
    return (uint8_t)(x >> (8 * (j & 3)));
}

// per byte of w: 0x80 where the byte is >= 64 (bit 6 or bit 7 set; what the shift brings in from the byte below lands on bit 0)
__device__ __forceinline__ uint32_t pf_ge64(uint32_t w) { return ((w << 1) | w) & 0x80808080u; }

// four words of 0x80-per-byte flags (16 bytes) -> one bit per byte, byte 0's lowest.  The flags of word q go to bit q of their byte,
// the four low nibbles are drawn together (bit 4 j + q for byte j of word q) and the 4 x 4 bit matrix is transposed by two delta
// swaps (bit 4 q + j): 20 plain operations -- a multiply per word to draw a word's flags together costs four issue slots each
__device__ __forceinline__ uint32_t pf_gather16(uint32_t f0, uint32_t f1, uint32_t f2, uint32_t f3)
{
    uint32_t v = (f0 >> 7) | (f1 >> 6) | (f2 >> 5) | (f3 >> 4);
    v = (v | (v >> 4)) & 0x00FF00FFu;
    v = (v | (v >> 8)) & 0xFFFFu;
    uint32_t t = ((v >> 3) ^ v) & 0x0A0Au;
    v ^= t ^ (t << 3);
    t = ((v >> 6) ^ v) & 0x00CCu;
    v ^= t ^ (t << 6);
    return v;
}

// the bytes of [p, p + 16) that lie inside a contig range -> 16-bit mask.  [cb, ce) is a range the caller knows (the one its tile
// starts in); only chunks that leave it walk the list.
__device__ __forceinline__ uint32_t pf_range_mask(const PackFastaArgs &a, uint64_t p, uint64_t cb, uint64_t ce, uint32_t r_first)
{
    if (p >= cb && p + 16 <= ce)
        return 0xFFFFu;
    uint32_t m = 0;
    uint32_t r = r_first;
    while (r < a.n_rng && a.rng_end[r] <= p)
        ++r;
    for (; r < a.n_rng && a.rng_begin[r] < p + 16; ++r) {
        const uint64_t b = a.rng_begin[r] > p ? a.rng_begin[r] - p : 0, e = a.rng_end[r] < p + 16 ? a.rng_end[r] - p : 16;
        if (e > b)
            m |= ((1u << e) - 1u) & ~((1u << b) - 1u);
    }
    return m;
}

__device__ __forceinline__ PfChunk pf_chunk(const PackFastaArgs &a, uint64_t p, uint64_t cb, uint64_t ce, uint32_t r_first, uint32_t w[4])
{
    PfChunk c;
    c.bits = 0;
    c.cnt = 0;
    c.keep = 0;
    c.other = 0;
    if (p >= a.n_raw) {
        w[0] = w[1] = w[2] = w[3] = 0;
        return c;
    }
    if (p + 16 <= a.n_raw) {
        const uint4 v = *(const uint4 *)(a.raw + p); // (p is a multiple of 16, the buffer 16-byte aligned)
        w[0] = v.x, w[1] = v.y, w[2] = v.z, w[3] = v.w;
    } else {
        w[0] = w[1] = w[2] = w[3] = 0;
#pragma unroll
        for (uint32_t j = 0; j < 16; ++j) // (static indices: the words stay in registers)
            if (p + j < a.n_raw)
                w[j >> 2] |= (uint32_t)a.raw[p + j] << (8 * (j & 3));
    }
    // ACGT / acgt: the code is bits 1..2 of the letter with G and T exchanged (A 0x41 C 0x43 G 0x47 T 0x54 -> 0 1 3 2 -> 0 1 2 3) --
    // a byte permute with the two bits as selector reads the code, and the letter the two bits stand for, out of a register
    uint32_t m[4], diffs[4], odd = 0, bad = 0;
    uint32_t bits;
    {
        uint32_t code[4];
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            const uint32_t x = w[q];
            m[q] = pf_ge64(x);                                            // 0x80 per byte >= 64
            const uint32_t t = (x >> 1) & 0x03030303u;
            code[q] = __builtin_amdgcn_perm(0u, 0x02030100u, t);          // t -> 0 1 3 2
            const uint32_t expect = __builtin_amdgcn_perm(0u, 0x47544341u, t); // t -> 'A' 'C' 'T' 'G'
            const uint32_t full = m[q] | (m[q] - (m[q] >> 7));            // 0xFF per byte >= 64: only those count
            diffs[q] = ((x & 0xDFDFDFDFu) ^ expect) & full;               // (case bit cleared)
            odd |= diffs[q];
        }
        // the sixteen 2-bit codes side by side, byte 0's lowest: word q's codes go to bit pair q of their byte (element 4 j + q),
        // then the 4 x 4 matrix of 2-bit elements is transposed (element 4 q + j) by two delta swaps
        uint32_t u = code[0] | (code[1] << 2) | (code[2] << 4) | (code[3] << 6);
        uint32_t t = ((u >> 6) ^ u) & 0x00CC00CCu;
        u ^= t ^ (t << 6);
        t = ((u >> 12) ^ u) & 0x0000F0F0u;
        u ^= t ^ (t << 12);
        bits = u;
    }
    uint32_t keep = pf_gather16(m[0], m[1], m[2], m[3]);
    // which bytes are outside ACGT is worked out only when some lane of the wavefront met one (N runs, IUPAC codes: rare)
    if (__ballot(odd != 0)) {
        uint32_t nz[4];
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q)
            nz[q] = (diffs[q] | ((diffs[q] & 0x7F7F7F7Fu) + 0x7F7F7F7Fu)) & 0x80808080u; // 0x80 per byte that differs
        bad = pf_gather16(nz[0], nz[1], nz[2], nz[3]);
    }
    keep &= pf_range_mask(a, p, cb, ce, r_first);
    bad &= keep;
    if (bad) { // a symbol outside ACGT: its code's two low bits stand in the word (nobody reads them: the block is escaped)
        for (uint32_t mm = bad; mm; mm &= mm - 1) {
            const uint32_t j = __builtin_ctz(mm);
            bits = (bits & ~(3u << (2 * j))) | ((uint32_t)(cnv_symbol(pf_byte(w, j)) & 3u) << (2 * j));
            c.other |= 1u << __popc(keep & ((1u << j) - 1u)); // (its rank among the kept symbols: what the staging clips by)
        }
    }
    // squeeze the dropped bytes' fields out, highest first
    for (uint32_t m = ~keep & 0xFFFFu; m;) {
        const uint32_t j = 31 - __builtin_clz(m);
        m &= ~(1u << j);
        const uint32_t low = j ? (0xFFFFFFFFu >> (32 - 2 * j)) : 0u;
        bits = (bits & low) | ((bits >> 2) & ~low);
    }
    c.bits = bits;
    c.keep = keep;
    c.cnt = __popc(keep);
    if (c.cnt < 16)
        c.bits &= c.cnt ? (0xFFFFFFFFu >> (32 - 2 * c.cnt)) : 0u;
    return c;
}

// inclusive scan of v over the 64 lanes
__device__ __forceinline__ uint32_t pf_wave_incl(uint32_t v)
{
    const uint32_t lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t u = __shfl_up(v, o);
        if (lane >= (uint32_t)o)
            v += u;
    }
    return v;
}

// symbols [g, g + cnt) with the 2-bit fields `bits` -> the tile's word stage in LDS, as far as they lie in [s0, s1)
__device__ __forceinline__ void pf_stage(uint32_t *lds_words, uint32_t *lds_flag, uint64_t g, uint32_t cnt, uint32_t bits, uint32_t other, uint64_t s0, uint64_t s1)
{
    if (!cnt || g + cnt <= s0 || g >= s1)
        return;
    if (g < s0) {
        const uint32_t d = (uint32_t)(s0 - g);
        bits >>= 2 * d;
        other >>= d;
        cnt -= d;
        g = s0;
    }
    if (g + cnt > s1) {
        cnt = (uint32_t)(s1 - g);
        bits &= 0xFFFFFFFFu >> (32 - 2 * cnt);
        other &= 0xFFFFu >> (16 - cnt);
    }
    const uint32_t li = (uint32_t)(g - s0), sh = 2 * (li & 15);
    atomicOr(&lds_words[li >> 4], bits << sh);
    if (sh && (li & 15) + cnt > 16)
        atomicOr(&lds_words[(li >> 4) + 1], bits >> (32 - sh));
    if (other) { // the block(s) of the symbols outside ACGT that are left after the clipping (a chunk spans two blocks at most)
        const uint32_t first = PACK_BLOCK - (li & (PACK_BLOCK - 1)); // symbols of this chunk that lie in the first of them
        if (first >= 16 || (other & ((1u << first) - 1u)))
            lds_flag[li / PACK_BLOCK] = 1;
        if (first < 16 && (other >> first))
            lds_flag[li / PACK_BLOCK + 1] = 1;
    }
}

// the escaped form of the same symbols: one byte each, cnv_num's codes
__device__ __forceinline__ void pf_escape(const PackFastaArgs &a, const int32_t *lds_slot, uint64_t g, uint32_t keep, const uint32_t w[4], uint64_t s0, uint64_t s1)
{
    for (uint32_t m = keep; m; m &= m - 1, ++g) {
        if (g < s0 || g >= s1)
            continue;
        const int32_t slot = lds_slot[(g - s0) / PACK_BLOCK];
        if (slot >= 0)
            a.esc_bytes[(uint64_t)slot * PACK_BLOCK + ((g - s0) & (PACK_BLOCK - 1))] = cnv_symbol(pf_byte(w, __builtin_ctz(m)));
    }
}

// symbols (bytes >= 64 inside a contig range) per tile
__global__ void __launch_bounds__(256) pack_fasta_count_kernel(PackFastaArgs a, uint32_t *__restrict__ tile_cnt)
{
    __shared__ uint32_t s_part[4], s_rfirst;
    __shared__ unsigned long long s_cb, s_ce;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, tile = blockIdx.x;
    const uint64_t t_begin = (uint64_t)tile * PF_TILE;
    if (tid == 0) {
        uint32_t lo = 0, hi = a.n_rng;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (a.rng_end[mid] <= t_begin)
                lo = mid + 1;
            else
                hi = mid;
        }
        s_rfirst = lo;
        s_cb = lo < a.n_rng ? a.rng_begin[lo] : ~0ULL;
        s_ce = lo < a.n_rng ? a.rng_end[lo] : 0;
    }
    __syncthreads();
    const uint32_t r_first = s_rfirst;
    const uint64_t cb = s_cb, ce = s_ce;
    uint32_t c = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
        const uint64_t p = t_begin + (uint64_t)wv * 4096 + j * 1024 + lane * 16;
        if (p >= a.n_raw)
            continue;
        uint32_t w[4] = {0, 0, 0, 0};
        if (p + 16 <= a.n_raw) {
            const uint4 v = *(const uint4 *)(a.raw + p);
            w[0] = v.x, w[1] = v.y, w[2] = v.z, w[3] = v.w;
        } else {
#pragma unroll
            for (uint32_t t = 0; t < 16; ++t)
                if (p + t < a.n_raw)
                    w[t >> 2] |= (uint32_t)a.raw[p + t] << (8 * (t & 3));
        }
        const uint32_t m0 = pf_ge64(w[0]), m1 = pf_ge64(w[1]), m2 = pf_ge64(w[2]), m3 = pf_ge64(w[3]);
        if (p >= cb && p + 16 <= ce) // (inside the range the tile starts in: nearly every chunk)
            c += __popc(m0) + __popc(m1) + __popc(m2) + __popc(m3);
        else
            c += __popc(pf_gather16(m0, m1, m2, m3) & pf_range_mask(a, p, cb, ce, r_first));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        c += __shfl_down(c, o);
    if (lane == 0)
        s_part[wv] = c;
    __syncthreads();
    if (tid == 0)
        tile_cnt[tile] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
}

// tile counts -> the symbols in front of every tile, in two levels (a single block walking 186 k counts took 0.44 ms -- as long as
// the counting pass itself): block b scans the counts of tiles [8192 b, 8192 (b + 1)) and leaves its total; the few totals are
// scanned by pp_scan_kernel; the pack kernel adds the two
constexpr uint32_t PF_SCAN_TILES = 8192; // tiles per scan block: 1024 threads x 8
__global__ void __launch_bounds__(1024) pack_fasta_scan_kernel(const uint32_t *__restrict__ tile_cnt, uint32_t n_tiles, uint32_t *__restrict__ tile_local,
                                                              uint32_t *__restrict__ block_total)
{
    __shared__ uint32_t s_w[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t t0 = blockIdx.x * PF_SCAN_TILES + tid * 8;
    uint32_t v[8], sum = 0;
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j) {
        v[j] = t0 + j < n_tiles ? tile_cnt[t0 + j] : 0u;
        sum += v[j];
    }
    const uint32_t incl = pf_wave_incl(sum);
    if (lane == 63)
        s_w[wv] = incl;
    __syncthreads();
    uint32_t base = 0, total = 0;
#pragma unroll
    for (uint32_t x = 0; x < 16; ++x) {
        if (x < wv)
            base += s_w[x];
        total += s_w[x];
    }
    uint32_t run = base + incl - sum;
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j) {
        if (t0 + j < n_tiles)
            tile_local[t0 + j] = run;
        run += v[j];
    }
    if (tid == 0)
        block_total[blockIdx.x] = total;
}

__global__ void __launch_bounds__(256) pack_fasta_kernel(PackFastaArgs a)
{
    __shared__ uint32_t s_words[PF_MAX_BLOCKS * (PACK_BLOCK / 16)];
    __shared__ uint32_t s_flag[PF_MAX_BLOCKS];
    __shared__ int32_t s_slot[PF_MAX_BLOCKS];
    __shared__ uint32_t s_wsum[4 * 4]; // [chunk row j][wave]
    __shared__ uint32_t s_tile, s_rfirst;
    __shared__ unsigned long long s_excl, s_cb, s_ce;
    __shared__ uint32_t s_ra[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;

    if (tid == 0)
        s_tile = blockIdx.x;
    for (uint32_t i = tid; i < PF_MAX_BLOCKS * (PACK_BLOCK / 16); i += 256)
        s_words[i] = 0;
    if (tid < PF_MAX_BLOCKS) {
        s_flag[tid] = 0;
        s_slot[tid] = -1;
    }
    __syncthreads();
    const uint32_t tile = s_tile;
    if (tile >= a.n_tiles)
        return;
    const uint64_t t_begin = (uint64_t)tile * PF_TILE, t_end = t_begin + PF_TILE < a.n_raw ? t_begin + PF_TILE : a.n_raw;
    if (tid == 0) {
        // the first range that ends behind the tile's first byte
        uint32_t lo = 0, hi = a.n_rng;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (a.rng_end[mid] <= t_begin)
                lo = mid + 1;
            else
                hi = mid;
        }
        s_rfirst = lo;
        s_cb = lo < a.n_rng ? a.rng_begin[lo] : ~0ULL;
        s_ce = lo < a.n_rng ? a.rng_end[lo] : 0;
    }
    __syncthreads();
    const uint32_t r_first = s_rfirst;
    const uint64_t cb = s_cb, ce = s_ce;

    // ---- own tile: four rows of 64 x 16 bytes per wave, every load coalesced
    uint32_t w[4][4];
    PfChunk ch[4];
    uint32_t rank[4]; // exclusive rank of the chunk's first symbol inside the tile
    uint32_t wave_tot = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
        const uint64_t p = t_begin + (uint64_t)wv * 4096 + j * 1024 + lane * 16;
        ch[j] = pf_chunk(a, p, cb, ce, r_first, w[j]);
    }
#pragma unroll
    for (uint32_t j = 0; j < 4; j += 2) { // (two rows per scan: a row's counts add up to 1024 at most, 16 bits each)
        const uint32_t incl = pf_wave_incl(ch[j].cnt | (ch[j + 1].cnt << 16));
        const uint32_t tot = (uint32_t)__shfl((int)incl, 63);
        rank[j] = wave_tot + (incl & 0xFFFFu) - ch[j].cnt;
        rank[j + 1] = wave_tot + (tot & 0xFFFFu) + (incl >> 16) - ch[j + 1].cnt;
        wave_tot += (tot & 0xFFFFu) + (tot >> 16);
    }
    if (lane == 0)
        s_wsum[wv] = wave_tot;
    __syncthreads();
    uint32_t wbase = 0, cnt_tile = 0;
#pragma unroll
    for (uint32_t x = 0; x < 4; ++x) {
        if (x < wv)
            wbase += s_wsum[x];
        cnt_tile += s_wsum[x];
    }

    // ---- the tile's offset: publish the count, look back.  The state word is all that travels between tiles (flag and value in one
    // 64-bit word), so the atomics are RELAXED: an agent-scope release / acquire pair would write back and invalidate the XCD's L2
    // around every one of them (buffer_wbl2 / buffer_inv on gfx950) -- measured: 19.7 ms per 3 Gbp instead of < 1
    if (tid == 0)
        s_excl = a.block_off[tile / PF_SCAN_TILES] + a.tile_local[tile];
    __syncthreads();
    const uint64_t o = s_excl;

    // ---- contigs that start in this tile: their symbol offset (the lane whose chunk holds the first byte)
    if (cb >= t_begin || (r_first + 1 < a.n_rng && a.rng_begin[r_first + 1] < t_end)) {
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const uint64_t p = t_begin + (uint64_t)wv * 4096 + j * 1024 + lane * 16;
            for (uint32_t r = r_first; r < a.n_rng && a.rng_begin[r] < p + 16; ++r)
                if (a.rng_begin[r] >= p && a.rng_begin[r] < a.n_raw)
                    a.ctg_off[r] = o + wbase + rank[j] + __popc(ch[j].keep & ((1u << (a.rng_begin[r] - p)) - 1u));
        }
    }
    if (tile == a.n_tiles - 1 && tid == 0)
        *a.total = o + cnt_tile;

    // ---- the blocks this tile owns: those whose first symbol it holds
    const uint64_t s0 = (o + PACK_BLOCK - 1) / PACK_BLOCK * PACK_BLOCK;
    if (s0 >= o + cnt_tile)
        return; // (none: its symbols belong to the last block of a tile in front)
    const uint64_t s1 = (o + cnt_tile - 1) / PACK_BLOCK * PACK_BLOCK + PACK_BLOCK;
    const uint32_t n_own = (uint32_t)((s1 - s0) / PACK_BLOCK);
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j)
        pf_stage(s_words, s_flag, o + wbase + rank[j], ch[j].cnt, ch[j].bits, ch[j].other, s0, s1);
    // ---- the rest of the last block, from the bytes behind the tile (rounds of 256 x 16 bytes)
    // (a round looks at as many bytes as the missing symbols need at a line width of 16 or more, not at 4 KiB: what lies behind the
    // tile is another tile's input, quite possibly in another XCD's L2)
    uint64_t g_next = o + cnt_tile, p_next = t_begin + PF_TILE;
    uint32_t n_rounds = 0;
    while (g_next < s1 && p_next < a.n_raw) {
        uint32_t wr[4];
        const uint32_t need = (uint32_t)(s1 - g_next), span = min(4096u, (need + need / 16 + 47) & ~15u);
        const PfChunk c = pf_chunk(a, tid * 16 < span ? p_next + tid * 16 : a.n_raw, cb, ce, r_first, wr);
        const uint32_t incl = pf_wave_incl(c.cnt);
        __syncthreads(); // (s_ra of the round before has been read)
        if (lane == 63)
            s_ra[wv] = incl;
        __syncthreads();
        uint32_t base = 0, tot = 0;
#pragma unroll
        for (uint32_t x = 0; x < 4; ++x) {
            if (x < wv)
                base += s_ra[x];
            tot += s_ra[x];
        }
        pf_stage(s_words, s_flag, g_next + base + incl - c.cnt, c.cnt, c.bits, c.other, s0, s1);
        g_next += tot;
        p_next += span;
        ++n_rounds;
    }
    __syncthreads();
    // ---- escaped blocks: a slot each
    if (tid < n_own) {
        int32_t slot = -1;
        if (s_flag[tid]) {
            const uint32_t s = atomicAdd(a.esc_count, 1u);
            slot = s < a.esc_cap ? (int32_t)s : -2; // (over capacity: the caller comes again with a larger buffer)
        }
        s_slot[tid] = slot;
        a.esc_index[s0 / PACK_BLOCK + tid] = slot;
    }
    // ---- the words
    {
        uint32_t *out = a.words + s0 / 16;
        const uint32_t n_words = n_own * (PACK_BLOCK / 16);
        for (uint32_t i = tid * 4; i < n_words; i += 1024)
            *(uint4 *)(out + i) = make_uint4(s_words[i], s_words[i + 1], s_words[i + 2], s_words[i + 3]); // (s0 / 16 is a multiple of 64 words)
    }
    __syncthreads();
    bool any = false;
    for (uint32_t i = 0; i < n_own; ++i)
        any |= s_slot[i] >= 0;
    if (!any)
        return;
    // ---- ... and their symbols one byte each: the same walk again (rare: N runs, IUPAC codes)
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j)
        pf_escape(a, s_slot, o + wbase + rank[j], ch[j].keep, w[j], s0, s1);
    g_next = o + cnt_tile;
    p_next = t_begin + PF_TILE;
    for (uint32_t r = 0; r < n_rounds; ++r) {
        uint32_t wr[4];
        const uint32_t need = (uint32_t)(s1 - g_next), span = min(4096u, (need + need / 16 + 47) & ~15u);
        const PfChunk c = pf_chunk(a, tid * 16 < span ? p_next + tid * 16 : a.n_raw, cb, ce, r_first, wr);
        const uint32_t incl = pf_wave_incl(c.cnt);
        __syncthreads();
        if (lane == 63)
            s_ra[wv] = incl;
        __syncthreads();
        uint32_t base = 0, tot = 0;
#pragma unroll
        for (uint32_t x = 0; x < 4; ++x) {
            if (x < wv)
                base += s_ra[x];
            tot += s_ra[x];
        }
        pf_escape(a, s_slot, g_next + base + incl - c.cnt, c.keep, wr, s0, s1);
        g_next += tot;
        p_next += span;
    }
    // (the tail of a last, partial block in an escaped slot stays as it is: nobody reads beyond n_symbols)
}

} // namespace agc
