// lz_kernels.hip -- LZ-diff index build and parse kernels for gfx950 (wave64).
//
// Reference behaviour reproduced bit-exactly (file:line under the reference tree):
//   index build  CLZDiffBase::prepare_index / make_index16/32   src/common/lz_diff.cpp:81-141, 375-428
//   best match   CLZDiffBase::find_best_match16/32              src/common/lz_diff.cpp:287-372
//   encode       CLZDiff_V2::Encode                             src/common/lz_diff.cpp:669-798
//   estimate     CLZDiff_V2::Estimate                           src/common/lz_diff.cpp:839-946
//   cost vector  CLZDiffBase::GetCodingCostVector               src/common/lz_diff.cpp:159-284
//
// Mapping: one wavefront parses one segment (the greedy parse is serial); the 64
// lanes are the 64 linear probes of a hash lookup (max_no_tries = 64), and the
// forward match extension compares 64 x 16 B per step with a ballot to find the
// first difference.  HBM-bound byte work: no MFMA anywhere.
#include "dev_common.h"

#ifndef AGC_TRACE
#define AGC_TRACE(code, val)
#endif
// -DAGC_PHASES (a profiling build, scripts/lz_phase_probe.sh): shader-clock time of the parse loop's phases, accumulated per wave
// and printed by a few waves of the encode launch.  Costs ~10 %; never part of the product build.
#ifdef AGC_PHASES
namespace agc {
__device__ unsigned long long g_phase_acc[8], g_phase_cnt[8];
}
#define PH_DECL uint64_t ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_last = __builtin_amdgcn_s_memtime();
#define PH(k)                                                  \
    {                                                          \
        const uint64_t ph_now = __builtin_amdgcn_s_memtime();  \
        ph_acc[k] += ph_now - ph_last;                         \
        ph_cnt[k] += 1;                                        \
        ph_last = ph_now;                                      \
    }
#else
#define PH_DECL
#define PH(k)
#endif

namespace agc {

// ---------------------------------------------------------------------------
// small wave helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & (WAVE - 1); }

__device__ __forceinline__ uint32_t ctz64(uint64_t x) { return (uint32_t)__builtin_ctzll(x); }

__device__ __forceinline__ uint32_t bcast_u32(uint32_t v, uint32_t src_lane)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)src_lane);
}

__device__ __forceinline__ uint32_t uniform_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

// spread the low 32 bits of x to the even bit positions of a 64-bit word
__device__ __forceinline__ uint64_t spread_bits(uint64_t x)
{
    x &= 0xffffffffULL;
    x = (x | (x << 16)) & 0x0000ffff0000ffffULL;
    x = (x | (x << 8)) & 0x00ff00ff00ff00ffULL;
    x = (x | (x << 4)) & 0x0f0f0f0f0f0f0f0fULL;
    x = (x | (x << 2)) & 0x3333333333333333ULL;
    x = (x | (x << 1)) & 0x5555555555555555ULL;
    return x;
}

// wave-uniform values the compiler cannot prove uniform (descriptor fields loaded through an index derived from threadIdx):
// kept in scalar registers explicitly
__device__ __forceinline__ uint64_t uniform_u64(uint64_t v)
{
    const uint32_t lo = uniform_u32((uint32_t)v), hi = uniform_u32((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
template <typename T> __device__ __forceinline__ T *uniform_ptr(T *p) { return (T *)uniform_u64((uint64_t)p); }
__device__ __forceinline__ SymView uniform_view(const SymView &v)
{
    SymView u;
    u.words = uniform_ptr(v.words);
    u.esc_index = uniform_ptr(v.esc_index);
    u.esc_bytes = uniform_ptr(v.esc_bytes);
    u.start = uniform_u64(v.start);
    u.len = uniform_u32(v.len);
    u.rc = uniform_u32(v.rc);
    return u;
}

// A view whose pointers name the global address space.  A pointer that reaches a kernel through a descriptor loaded from memory is a
// generic pointer to the compiler: every access becomes a FLAT instruction, which counts against the LDS / scalar-memory counter as
// well -- a read of the text window in LDS then waits for every table row and reference word still in flight.  The parse converts
// its views once (the sym_view.h accessors are templates over the view type).
typedef const __attribute__((address_space(1))) uint32_t g_u32;
typedef const __attribute__((address_space(1))) int32_t g_i32;
typedef const __attribute__((address_space(1))) uint8_t g_u8;
typedef const __attribute__((address_space(1))) uint64_t g_u64;
typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));
typedef uint64_t v2u64 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) __attribute__((aligned(4))) v4u32 g_v4u32_a4; // (16 bytes at a 4-byte aligned address)
typedef __attribute__((address_space(1))) __attribute__((aligned(8))) v2u64 g_v2u64_a8;
struct SymViewG {
    g_u32 *words;
    g_i32 *esc_index;
    g_u8 *esc_bytes;
    uint64_t start;
    uint32_t len, rc;
};
__device__ __forceinline__ SymViewG global_view(const SymView &v)
{
    return {(g_u32 *)v.words, (g_i32 *)v.esc_index, (g_u8 *)v.esc_bytes, v.start, v.len, v.rc};
}

// no block the sequence touches is escaped (whole wave; one index entry per lane and step)
template <class V> __device__ bool wave_view_clean(const V &v)
{
    if (!v.esc_index || !v.len)
        return true;
    const uint64_t b0 = v.start / SV_BLOCK, b1 = (v.start + v.len - 1) / SV_BLOCK;
    bool any = false;
    for (uint64_t b = b0 + lane_id(); b <= b1; b += WAVE)
        any = any || v.esc_index[b] >= 0;
    return __ballot(any) == 0;
}

// a lane's chunk (N = 16 or 32 symbols: P, I as sv_fetch delivers them; cnt of them valid) as bytes at dst (LDS, N-byte aligned)
template <uint32_t N, class V>
__device__ __forceinline__ void store_syms(uint8_t *dst, const V &v, uint32_t pos, uint32_t cnt, uint64_t P, uint32_t I)
{
    if (cnt == N && I == 0) {
        const uint32_t lo = (uint32_t)P, hi = (uint32_t)(P >> 32);
        *(uint4 *)dst = make_uint4(sv_expand4(lo), sv_expand4(lo >> 8), sv_expand4(lo >> 16), sv_expand4(lo >> 24));
        if (N == 32)
            *(uint4 *)(dst + 16) = make_uint4(sv_expand4(hi), sv_expand4(hi >> 8), sv_expand4(hi >> 16), sv_expand4(hi >> 24));
    } else {
        for (uint32_t j = 0; j < cnt; ++j)
            dst[j] = (uint8_t)(((I >> j) & 1u) ? sv_sym(v, pos + j) : (uint32_t)(P >> (2u * j)) & 3u);
    }
}

// Per-wave window of the text in LDS, one byte per symbol: the symbols the parse is about to look at (keys of the next
// positions) without another trip to HBM.  It is refilled from the packed words when the parse position leaves it (512 bytes
// of HBM for 2048 symbols) and -- for free -- by the last step of every forward compare, whose text chunk contains the
// position right after the match.
#ifndef AGC_LZ_CMP
#define AGC_LZ_CMP 16
#endif
// symbols a lane compares per step: 16 (two dwords per side) or 32 (three).  Measured on the 3 Gbp step: the same kernel time
// (4.2 ms) -- a second step for the 37 % of matches beyond 1024 symbols costs what the narrower loads save -- and a third less
// HBM traffic with 16.
constexpr uint32_t CMP_SYMS = AGC_LZ_CMP;
constexpr uint32_t WIN_SYMS = WAVE * CMP_SYMS; // 1024 / 2048
struct TextWin {
    uint8_t *lds;   // WIN_SYMS bytes of LDS owned by this wave
    uint32_t base;  // text position of lds[0]
    uint32_t len;   // valid symbols
};

__device__ __forceinline__ bool win_has(const TextWin &w, uint32_t pos, uint32_t cnt)
{
    return pos >= w.base && pos + cnt <= w.base + w.len;
}

template <class V> __device__ __forceinline__ void win_fill(TextWin &w, const V &tv, bool t_clean, uint32_t n, uint32_t pos)
{
    const uint32_t lane = lane_id();
    const uint32_t len = n - pos < WIN_SYMS ? n - pos : WIN_SYMS;
    const uint32_t off = lane * CMP_SYMS;
    if (off < len) {
        const uint32_t cnt = len - off < CMP_SYMS ? len - off : CMP_SYMS;
        uint64_t P;
        uint32_t I;
        sv_fetch<CMP_SYMS>(tv, pos + off, cnt, t_clean, P, I);
        store_syms<CMP_SYMS>(w.lds + off, tv, pos + off, cnt, P, I);
    }
    w.base = pos;
    w.len = len;
    __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0): the LDS stores are done before the wave reads them back
}

// Length of the common prefix of text[tp ..) and ref[rp ..), at most max_len symbols (the caller bounds it by both ends),
// whole wave: a lane takes 32 symbols of both sides per step (three dwords each) and finds its first difference with one
// 64-bit XOR; symbols outside ACGT (escaped blocks) are compared by value, one at a time.
// Semantics of refresh::matching_length (3rd_party/refresh/string_operations/lib/string_operations.h:18-69) as used by
// compare_fwd (lz_diff.h:264-266).  When `win` is given the text chunk of the final step is captured into the window.
template <class V>
__device__ uint32_t wave_match_fwd(const V &tv, uint32_t tp, bool t_clean, const V &rv, uint32_t rp, bool r_clean, uint32_t max_len,
                                   TextWin *win = nullptr)
{
    const uint32_t lane = lane_id();
    for (uint32_t base = 0;; base += WIN_SYMS) {
        const uint32_t off = base + lane * CMP_SYMS;
        bool stop = true;
        uint32_t so = 0, cnt = 0, It = 0;
        uint64_t Pt = 0;
        if (off < max_len) {
            cnt = max_len - off < CMP_SYMS ? max_len - off : CMP_SYMS;
            uint64_t Pr;
            uint32_t Ir;
            sv_fetch<CMP_SYMS>(tv, tp + off, cnt, t_clean, Pt, It);
            sv_fetch<CMP_SYMS>(rv, rp + off, cnt, r_clean, Pr, Ir);
            const uint64_t d = CMP_SYMS == 32 ? Pt ^ Pr : (uint64_t)(uint32_t)(Pt ^ Pr);
            uint32_t j = d ? sv_ctz64(d) >> 1 : CMP_SYMS;
            const uint32_t inv = It | Ir;
            if (inv) {
                const uint32_t ji = (uint32_t)__builtin_ctz(inv);
                if (ji < j) { // from the first symbol outside ACGT on: by value
                    j = ji;
                    while (j < cnt && sv_sym(tv, tp + off + j) == sv_sym(rv, rp + off + j))
                        ++j;
                }
            }
            so = j < cnt ? j : cnt;
            stop = so < CMP_SYMS; // (a chunk cut by max_len stops too)
        }
        const uint64_t m = __ballot(stop);
        if (m) {
            if (win) {
                // lanes holding a full chunk form a prefix of the wave: capture them
                const bool full = cnt == CMP_SYMS;
                const uint64_t fm = __ballot(full);
                if (full)
                    store_syms<CMP_SYMS>(win->lds + lane * CMP_SYMS, tv, tp + off, CMP_SYMS, Pt, It);
                win->base = tp + base;
                win->len = (uint32_t)__builtin_popcountll(fm) * CMP_SYMS;
                __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0)
            }
            const uint32_t l = ctz64(m);
            return base + l * CMP_SYMS + bcast_u32(so, l);
        }
    }
}

// Backward extension (lz_diff.cpp:308-311): number of equal symbols walking left from text[tp - 1] / ref[rp - 1], at most lim.
template <class V> __device__ uint32_t wave_common_suffix(const V &tv, uint32_t tp, const V &rv, uint32_t rp, uint32_t lim)
{
    const uint32_t lane = lane_id();
    for (uint32_t base = 0; base < lim; base += WAVE) {
        const uint32_t idx = base + lane;
        bool mism = true;
        if (idx < lim)
            mism = sv_sym(tv, tp - 1 - idx) != sv_sym(rv, rp - 1 - idx);
        const uint64_t m = __ballot(mism);
        if (m)
            return base + ctz64(m);
    }
    return lim;
}

// decimal length / emission (append_int, lz_diff.h:229-262)
__device__ __forceinline__ uint32_t dec_len_u32(uint32_t x)
{
    uint32_t n = 1;
    while (x >= 10) {
        x /= 10;
        ++n;
    }
    return n;
}

__device__ __forceinline__ uint32_t base_dec_len(uint32_t x)
{
    if (x < 10) return 1;
    if (x < 100) return 2;
    if (x < 1000) return 3;
    if (x < 10000) return 4;
    if (x < 100000) return 5;
    if (x < 1000000) return 6;
    if (x < 10000000) return 7;
    if (x < 100000000) return 8;
    if (x < 1000000000) return 9;
    return 10;
}

// writes the decimal form of x (with '-' when negative) at out[o..); every lane computes
// the returned length, only `writer` stores.  32-bit arithmetic only: the values are
// differences of u32 positions taken as int (lz_diff.cpp:633) and u32 lengths, and a 64-bit
// integer division costs hundreds of VALU instructions on this hardware.
__device__ __forceinline__ uint32_t emit_uint(uint8_t *out, uint32_t o, uint32_t ax, bool writer)
{
    const uint32_t nd = base_dec_len(ax);
    if (writer) {
        uint32_t t = ax;
        for (uint32_t d = 0; d < nd; ++d) {
            const uint32_t q = t / 10u; // constant divisor: mul-hi + shift
            out[o + nd - 1 - d] = (uint8_t)('0' + (t - q * 10u));
            t = q;
        }
    }
    return nd;
}

__device__ __forceinline__ uint32_t emit_int(uint8_t *out, uint32_t o, int32_t x, bool writer)
{
    if (x < 0) {
        if (writer)
            out[o] = '-';
        return 1 + emit_uint(out, o + 1, (uint32_t)(-(int64_t)x), writer);
    }
    return emit_uint(out, o, (uint32_t)x, writer);
}

// cost helpers ---------------------------------------------------------------
// CLZDiff_V2::uint_len / int_len / cost_match / cost_Nrun, lz_diff.h:375-424
__device__ __forceinline__ uint32_t v2_uint_len(uint32_t x)
{
    if (x < 10) return 1;
    if (x < 100) return 2;
    if (x < 1000) return 3;
    if (x < 10000) return 4;
    if (x < 100000) return 5;
    if (x < 1000000) return 6;
    if (x < 10000000) return 7;
    return 8;
}
__device__ __forceinline__ uint32_t v2_cost_match(uint32_t mml, uint32_t ref_pos, uint32_t len, uint32_t pred_pos)
{
    int dif = (int)ref_pos - (int)pred_pos;
    uint32_t r = dif >= 0 ? v2_uint_len((uint32_t)dif) : 1 + v2_uint_len((uint32_t)-dif);
    if (len != ~0u)
        r += 1 + v2_uint_len(len - mml);
    return r + 1;
}
// CLZDiffBase::int_len / coding_cost_match / coding_cost_Nrun, lz_diff.h:159-191
__device__ __forceinline__ uint32_t base_int_len(uint32_t x)
{
    if (x < 10) return 1;
    if (x < 100) return 2;
    if (x < 1000) return 3;
    if (x < 10000) return 4;
    if (x < 100000) return 5;
    if (x < 1000000) return 6;
    if (x < 10000000) return 7;
    if (x < 100000000) return 8;
    if (x < 1000000000) return 9;
    return 10;
}
__device__ __forceinline__ uint32_t base_cost_match(uint32_t mml, uint32_t ref_pos, uint32_t len, uint32_t pred_pos)
{
    int dif = (int)ref_pos - (int)pred_pos;
    uint32_t r = dif >= 0 ? base_int_len((uint32_t)dif) : base_int_len((uint32_t)-dif) + 1;
    return r + base_int_len(len - mml) + 2;
}

// ---------------------------------------------------------------------------
// The parse, one wavefront per segment.
// ---------------------------------------------------------------------------
enum { MODE_ENCODE = 0, MODE_ESTIMATE = 1, MODE_COSTVEC = 2 };
// a per-position coding cost (GetCodingCostVector, lz_diff.cpp:159-284): 0 inside a run, 1 per literal, the token's length
// (a few decimal digits and separators, <= 2 * 10 + 2) at one end of a match or N run -- ONE BYTE in HBM: the vectors are written
// once and read by split_point_kernel; as 32-bit words they were 1.4 GB written + 2.1 GB read per 3 Gbp sample (round 2)
typedef uint8_t cost_t;

struct ParseOut {
    uint32_t value; // encode: delta length; estimate: cost; cost vector: number of costs
    uint32_t peak;  // estimate: largest cost seen at a loop-top check
};

// ---- a parse in chunks (round 5) ----
// One wavefront per text is a chain of dependent steps: ~7 us per edit, so a 500 kb text at 2.5 % divergence keeps ONE wavefront busy
// for 90 ms however empty the GPU is -- the shape of adaptive collections of diverged genomes, where a launch holds a few dozen long
// texts.  The state of the reference's loop at the top of an iteration is (i, pred_pos, no_prev_literals) (lz_diff.cpp:669-798), and
// right after a match it is (i, pred_pos, 0): nothing in front of a match end is read or re-written again (back extension and the
// '!' patch stop at the literals since the last match).  So a parser that starts somewhere in the text with a guessed state produces
// the reference's tokens from the first match end on that it shares with the true parse -- same position, same pred_pos -- and in
// similar sequences that is its first or second match.
//   ROLE 1 (lz_chunk_kernel): one wavefront per CHUNK of a text, started at the chunk's first symbol with pred_pos = 0; it stops at the
//          chunk's end and logs the state after every match (position, pred_pos, bytes written, running cost).
//   ROLE 2 (lz_hop_kernel): one wavefront per text walks from chunk to chunk: it is the true parse, but wherever its state after a match
//          is in the log of the chunk it has reached, it takes that chunk's tokens from there to the chunk's last match end as they are
//          (copies the bytes / adds the costs) and goes on behind them -- it parses only the few tokens around every chunk boundary.
//   ROLE 0: the whole text by one wavefront, as before (launches with thousands of texts: the GPU is full anyway).
struct ChunkState {
    uint32_t i, pred, o, est; // after a match: text position, pred_pos, output cursor (estimate: the peak), running cost
};
struct ChunkCtl {
    uint32_t i0, i_stop;          // ROLE 1: the chunk [i0, i_stop)
    ChunkState *log;              // ROLE 1: this chunk's log; ROLE 2: the logs of the text's chunks, `cap` entries apart
    uint32_t *log_n;              // ... and their entry counts
    uint32_t cap, chunk_len, n_chunks;
    const uint8_t *chunk_out;     // ROLE 2, encode: chunk c's bytes at chunk_out + c * chunk_stride
    uint32_t chunk_stride;
};

// maybe: one bit per text position from key_filter_kernel (0 = the key at this position is valid and not in the reference's
// index: a certain literal); nullptr: literal runs are found by probing the table (wide probe)
template <int MODE, int ROLE = 0>
__device__ ParseOut lz_parse(const RefDesc &rd, const SymViewG &tv, uint8_t *__restrict__ out, cost_t *__restrict__ costs,
                             const bool prefix_costs, uint8_t *win_lds, const unsigned long long *__restrict__ maybe_generic,
                             const ChunkCtl cc = ChunkCtl())
{
    const uint32_t lane = lane_id();
    const bool writer = lane == 0;
    const uint32_t n = tv.len;
    const uint32_t key_len = rd.key_len;
    const uint32_t mml = rd.min_match_len;
    const uint32_t ref_size = rd.ref_size;
    const uint32_t ht_mask = rd.ht_mask;
    const SymViewG rv = {(g_u32 *)rd.words, (g_i32 *)rd.esc_index, (g_u8 *)rd.esc_bytes, 0, ref_size, 0};
    g_u64 *maybe = (g_u64 *)maybe_generic;
    g_u32 *tab32 = (g_u32 *)rd.table;
    g_u64 *tab64 = (g_u64 *)rd.table;
    const bool r_clean = rd.esc_index == nullptr; // (a reference has an escape index only when it holds a symbol outside ACGT)
    const bool t_clean = wave_view_clean(tv);
    ParseOut res{0, 0};

    if (MODE != MODE_COSTVEC && ROLE != 1) {
        // identical sequence (lz_diff.cpp:678-680 / 849-851)
        if (n == ref_size && wave_match_fwd(tv, 0, t_clean, rv, 0, r_clean, n) == n)
            return res;
    }

    uint32_t i = ROLE == 1 ? cc.i0 : 0, pred_pos = 0, npl = 0; // npl = no_prev_literals
    uint32_t o = ROLE == 1 && MODE == MODE_COSTVEC ? cc.i0 : 0; // output cursor (bytes or costs; a cost's index is its position)
    uint32_t est = 0, peak = 0;
    const uint32_t i_stop = ROLE == 1 ? cc.i_stop : 0xFFFFFFFFu; // a chunk parser writes no cost at or beyond its chunk's end
    uint32_t n_log = 0;                                          // ROLE 1: entries logged
    uint32_t dbg_hops = 0, dbg_own = 0;                          // ROLE 2: chunks taken over / matches parsed by this wavefront itself
#define PUT_COST(idx, v)                                                                           \
    do {                                                                                           \
        if (ROLE != 1 || (idx) < i_stop)                                                           \
            costs[(idx)] = (v);                                                                    \
    } while (0)

    AGC_TRACE(3, n);
    const uint64_t keybits = (1ULL << key_len) - 1ULL;
    bool stale_out = false;   // bytes at/after `o` were stored by lane 0 earlier (rolled-back literals)
    bool coop_out = false;    // bytes before `o` were stored by lanes other than 0 since the last drain
    // Wide (64-position) probing pays off in long literal runs but costs one table line per position;
    // in the match / SNP / match rhythm of similar sequences the 1-4 literals after a mismatch are
    // cheaper as exact steps.  It is therefore armed only after WIDE_AFTER consecutive literal steps.
    constexpr uint32_t WIDE_AFTER = 4;
    // slots a lane may inspect in the wide probe (two 16-byte loads); a longer chain makes the
    // position a conservative "stop" that the exact step resolves -- the result is unchanged
    constexpr uint32_t WIDE_MAX_SLOTS = MAX_NO_TRIES; // (a budget of 8 made literal runs slower: more exact steps)
    uint32_t lit_streak = 0;
    bool try_wide = false;
    bool force_exact = false; // the grouped probe could not settle position i: the next step is the exact one
    TextWin win{win_lds, 0, 0};
    // ROLE 2: chunk x's tokens from its logged state `from` (== this parser's state) to its last logged state are taken as they are
    auto hop = [&](uint32_t x, const ChunkState from) {
        const ChunkState last = cc.log[(size_t)x * cc.cap + cc.log_n[x] - 1];
        if (MODE == MODE_ENCODE) {
            const uint8_t *src = cc.chunk_out + (size_t)x * cc.chunk_stride + from.o;
            const uint32_t cnt = last.o - from.o;
            if (stale_out) {
                __builtin_amdgcn_s_waitcnt(0); // lane 0's rolled-back bytes land before the copied chunk bytes overwrite them
                stale_out = false;
            }
            for (uint32_t t = lane; t < cnt; t += WAVE)
                out[o + t] = src[t];
            o += cnt;
            coop_out = true;
        } else if (MODE == MODE_ESTIMATE) {
            const uint32_t base = est - from.est;
            if (base + last.o > peak)
                peak = base + last.o; // (.o of an estimate's log entry: the chunk parser's peak at that point)
            est = base + last.est;
        } else
            o = last.i; // (the chunk parser wrote its costs where they belong)
        i = last.i;
        pred_pos = last.pred;
        npl = 0;
        lit_streak = 0;
        try_wide = false;
        force_exact = false;
    };
    // ROLE 2, after a match: is this state (position, pred_pos) in the log of the chunk the position lies in?
    auto try_hop = [&]() {
        if (!i)
            return;
        const uint32_t x = (i - 1) / cc.chunk_len;
        if (x >= cc.n_chunks)
            return;
        const uint32_t nl = cc.log_n[x];
        for (uint32_t b0 = 0; b0 < nl; b0 += WAVE) {
            ChunkState e{0xFFFFFFFFu, 0, 0, 0};
            if (b0 + lane < nl)
                e = cc.log[(size_t)x * cc.cap + b0 + lane];
            const uint64_t hit = __ballot(e.i == i && e.pred == pred_pos);
            if (hit) {
                const uint32_t l = ctz64(hit);
                if (b0 + l + 1 < nl) { // (the chunk's last entry: nothing behind it to take)
                    ChunkState from;
                    from.i = i;
                    from.pred = pred_pos;
                    from.o = bcast_u32(e.o, l);
                    from.est = bcast_u32(e.est, l);
                    hop(x, from);
                    ++dbg_hops;
                }
                return;
            }
            if (__ballot(b0 + lane < nl && e.i > i))
                return; // (entries ascend by position)
        }
    };
    if (ROLE == 2 && cc.n_chunks && cc.log_n[0])
        hop(0, ChunkState{0, 0, 0, 0}); // the first chunk's parser started with the true state
#define LOG_STATE()                                                                                \
    do {                                                                                           \
        if (ROLE == 1 && i <= i_stop && n_log < cc.cap) {                                          \
            if (writer)                                                                            \
                cc.log[n_log] = ChunkState{i, pred_pos, MODE == MODE_ESTIMATE ? peak : o, est};    \
            ++n_log;                                                                               \
        }                                                                                          \
        if (ROLE == 2) {                                                                           \
            ++dbg_own;                                                                             \
            try_hop();                                                                             \
        }                                                                                          \
    } while (0)
    PH_DECL
    while (i + key_len < n && (ROLE != 1 || i < i_stop)) {
        AGC_TRACE(4, i);
        PH(0) // everything after the previous iteration's last mark (emission of a match / literal)
        {
            // symbols i .. i+64+key_len+2 (as far as the text goes) must be in the LDS window
            const uint32_t need = n - i < WAVE + key_len + 2 ? n - i : WAVE + key_len + 2;
            if (!win_has(win, i, need))
                win_fill(win, tv, t_clean, n, i);
        }
        PH(1) // window check / refill
        const uint8_t *__restrict__ wtext = win.lds - win.base; // wtext[pos] for positions inside the window
        // ---- wide literal probe: lanes look at positions i .. i+63 at once.  A position is a
        // certain literal when its key is valid and no slot of its probe chain (up to the first
        // empty slot / 64 tries) carries the key's fingerprint, or when its key is invalid and no
        // N-run starts there.  The leading run of certain literals is emitted in one step; the
        // first other position is handled by the exact (reference-order) step below.
        // Only entered after an exact step found no match (long matches never pay for it).
        if (MODE != MODE_ENCODE && try_wide && maybe) {
            // literal run by the filter bitmap: the next position that may match, up to 4096 positions ahead (one 8-byte load
            // per lane); everything before it is a certain literal
            const uint32_t w0i = i >> 6;
            const uint32_t n_words = (n + 63) >> 6;
            uint64_t mw = w0i + lane < n_words ? maybe[w0i + lane] : ~0ULL; // past the text: "stop"
            if (lane == 0)
                mw &= ~0ULL << (i & 63);
            const uint64_t any = __ballot(mw != 0);
            const uint32_t l = ctz64(any);                 // (lane w0i + lane >= n_words always votes)
            const uint32_t first = (w0i + l) * 64 + (uint32_t)__builtin_ctzll(__shfl(mw, (int)l));
            uint32_t stop_pos = any ? first : (w0i + WAVE) * 64;
            // positions must satisfy q + key_len < n (the loop condition of the exact steps)
            const uint32_t lim = n - key_len; // i + key_len < n holds here, so lim > i
            if (stop_pos > lim)
                stop_pos = lim;
            if (ROLE == 1 && stop_pos > i_stop)
                stop_pos = i_stop; // (i < i_stop here: the loop condition)
            const uint32_t f = stop_pos - i;
            if (f) {
                if (MODE == MODE_ESTIMATE) {
                    if (est + f - 1 > peak)
                        peak = est + f - 1; // loop-top checks of these f literal steps
                    est += f;
                } else {
                    for (uint32_t t = lane; t < f; t += WAVE)
                        PUT_COST(o + t, 1);
                }
                o += f;
                i += f;
                pred_pos += f;
                npl += f;
                try_wide = !any && stop_pos < lim; // nothing found in this stretch: look at the next one
                if (!try_wide)
                    lit_streak = WIDE_AFTER; // the stop position gets its exact step; a literal there re-arms the skipping
                continue;
            }
            try_wide = false;
        } else if (try_wide) {
            const uint32_t q = i + lane;
            const uint32_t sa = q < n ? (uint32_t)wtext[q] : 0xFFu;
            const uint32_t sb = (lane < key_len + 2 && q + 64 < n) ? (uint32_t)wtext[q + 64] : 0xFFu;
            const uint64_t a0 = __ballot((sa & 1u) != 0), a1 = __ballot((sa & 2u) != 0), ai = __ballot(sa > 3), an = __ballot(sa == N_CODE);
            const uint64_t b0 = __ballot((sb & 1u) != 0), b1 = __ballot((sb & 2u) != 0), bi = __ballot(sb > 3), bn = __ballot(sb == N_CODE);
            const uint32_t sh = lane, rs = (64 - lane) & 63;
            const uint64_t hi_on = lane ? ~0ULL : 0ULL; // (x << 64) is undefined: lane 0 takes nothing from the second plane
            const uint64_t w0 = (a0 >> sh) | ((b0 << rs) & hi_on), w1 = (a1 >> sh) | ((b1 << rs) & hi_on);
            const uint64_t wi = (ai >> sh) | ((bi << rs) & hi_on), wn = (an >> sh) | ((bn << rs) & hi_on);
            bool stop;
            if (!(q + key_len < n))
                stop = true;
            else if (wi & keybits)
                stop = (wn & 7ULL) == 7ULL;
            else {
                const uint64_t r0 = __brevll(w0 & keybits) >> (64 - key_len), r1 = __brevll(w1 & keybits) >> (64 - key_len);
                const uint64_t hx = murmur64(spread_bits(r0) | (spread_bits(r1) << 1));
                uint32_t sl = (uint32_t)hx & ht_mask;
                stop = false;
                // walk the probe chain 16 bytes at a time (4 short / 2 long entries per load); chains
                // end at the first empty slot, so one or two round trips settle almost every lane
                if (rd.is_short) {
                    const uint32_t fp = (uint32_t)(hx >> 48);
                    g_u32 *tab = tab32;
                    bool done = false;
                    for (uint32_t t = 0; t < WIDE_MAX_SLOTS && !done; t += 4) {
                        uint32_t e[4];
                        if (sl + 3 <= ht_mask) {
                            const v4u32 v = *(const g_v4u32_a4 *)(tab + sl);
                            e[0] = v.x, e[1] = v.y, e[2] = v.z, e[3] = v.w;
                        } else {
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                e[u] = tab[(sl + u) & ht_mask];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if (!done) {
                                if (e[u] == 0xFFFFFFFFu)
                                    done = true;
                                else if ((e[u] & 0xFFFFu) == fp) {
                                    stop = true;
                                    done = true;
                                }
                            }
                        }
                        sl = (sl + 4) & ht_mask;
                    }
                    stop = stop || !done; // chain longer than the probe budget: let the exact step decide
                } else {
                    const uint32_t fp = (uint32_t)(hx >> 32);
                    g_u64 *tab = tab64;
                    bool done = false;
                    for (uint32_t t = 0; t < WIDE_MAX_SLOTS && !done; t += 2) {
                        uint64_t e[2];
                        if (sl + 1 <= ht_mask) {
                            const v2u64 v = *(const g_v2u64_a8 *)(tab + sl);
                            e[0] = v.x, e[1] = v.y;
                        } else {
                            e[0] = tab[sl & ht_mask];
                            e[1] = tab[(sl + 1) & ht_mask];
                        }
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            if (!done) {
                                if (e[u] == ~0ULL)
                                    done = true;
                                else if ((uint32_t)e[u] == fp) {
                                    stop = true;
                                    done = true;
                                }
                            }
                        }
                        sl = (sl + 2) & ht_mask;
                    }
                    stop = stop || !done;
                }
            }
            const uint64_t sm = __ballot(stop);
            uint32_t f = sm ? ctz64(sm) : WAVE;
            if (ROLE == 1 && f > i_stop - i)
                f = i_stop - i;
            if (f) {
                if (MODE == MODE_ENCODE) {
                    if (stale_out) {
                        __builtin_amdgcn_s_waitcnt(0); // lane 0's rolled-back bytes land before other lanes overwrite them
                        stale_out = false;
                    }
                    if (lane < f)
                        out[o + lane] = (uint8_t)('A' + sa);
                    coop_out = true;
                } else if (MODE == MODE_ESTIMATE) {
                    if (est + f - 1 > peak)
                        peak = est + f - 1; // loop-top checks of these f literal steps
                    est += f;
                } else {
                    // (no drain: positions at or after `o` hold no pending store -- a roll-back is always followed by a match
                    // that covers the rolled-back positions and drains around its own stores)
                    if (lane < f)
                        PUT_COST(o + lane, 1);
                }
                o += f;
                i += f;
                pred_pos += f;
                npl += f;
                try_wide = f == WAVE; // a stop position was seen: go straight to its exact step
                continue;
            }
            try_wide = false;
        }
        // ---- grouped probe: positions i .. i+3, 16 lanes = 16 probe slots each, one round trip.  In the match / SNP / match
        // rhythm the literal at the SNP and the 0-3 un-indexed positions after it (HASHING_STEP = 4) are settled together with
        // the probe of the first position that has candidates; what the exact step would have seen for that position (the
        // fingerprint hits before the first empty slot, in slot order) is handed to the verification below.  A group whose
        // chain does not end within 16 slots, or whose key holds a non-ACGT symbol, is left to the exact step.
        uint64_t cand = 0;
        uint32_t epos = 0;
        bool have_cands = false;
        if (!force_exact) {
            constexpr uint32_t MP_G = 4, MP_S = WAVE / MP_G;
            const uint32_t g = lane / MP_S;
            const uint32_t sa = (lane < key_len + MP_G - 1 && i + lane < n) ? (uint32_t)wtext[i + lane] : 0xFFu;
            const uint64_t a0 = __ballot((sa & 1u) != 0), a1 = __ballot((sa & 2u) != 0), ai = __ballot(sa > 3);
            const bool q_ok = i + g + key_len < n && ((ai >> g) & keybits) == 0;
            bool is_empty = false, fp_ok = false;
            if (q_ok) {
                const uint64_t r0 = __brevll((a0 >> g) & keybits) >> (64 - key_len), r1 = __brevll((a1 >> g) & keybits) >> (64 - key_len);
                const uint64_t hx = murmur64(spread_bits(r0) | (spread_bits(r1) << 1));
                const uint32_t sl = ((uint32_t)hx + (lane % MP_S)) & ht_mask;
                if (rd.is_short) {
                    const uint32_t e = tab32[sl];
                    is_empty = e == 0xFFFFFFFFu;
                    epos = e >> 16;
                    fp_ok = (e & 0xFFFFu) == (uint32_t)(hx >> 48);
                } else {
                    const uint64_t e = tab64[sl];
                    is_empty = e == ~0ULL;
                    epos = (uint32_t)(e >> 32);
                    fp_ok = (uint32_t)e == (uint32_t)(hx >> 32);
                }
            }
            const uint64_t em = __ballot(is_empty), fm = __ballot(fp_ok && !is_empty);
            PH(2) // grouped probe: keys, hashes, table rows (+ the speculative chunk loads) until the rows are in
            uint32_t f = 0;
            bool to_exact = false;
            for (; f < MP_G; ++f) {
                if (!(i + f + key_len < n))
                    break; // the loop ends here
                const uint32_t em16 = (uint32_t)(em >> (f * MP_S)) & 0xFFFFu;
                if (((ai >> f) & keybits) != 0 || !em16) {
                    to_exact = true;
                    break;
                }
                const uint32_t c16 = (uint32_t)(fm >> (f * MP_S)) & ((1u << __builtin_ctz(em16)) - 1u);
                if (c16) {
                    cand = (uint64_t)c16 << (f * MP_S);
                    have_cands = true;
                    break;
                }
            }
            if (f) {
                if (MODE == MODE_ENCODE) {
                    // (lane 0 owns these bytes like every token of the match / SNP rhythm: stores of one lane to one address
                    // keep their order, so nothing has to be drained when a match rolls literals back or patches them)
                    for (uint32_t t = 0; t < f; ++t) {
                        const uint32_t c = bcast_u32(sa, t);
                        if (writer)
                            out[o + t] = (uint8_t)('A' + c);
                    }
                } else if (MODE == MODE_ESTIMATE) {
                    if (est + f - 1 > peak)
                        peak = est + f - 1; // loop-top checks of these f literal steps
                    est += f;
                } else if (lane < f)
                    PUT_COST(o + lane, 1);
                o += f;
                i += f;
                pred_pos += f;
                npl += f;
                lit_streak += f;
            }
            if (!have_cands) {
                force_exact = to_exact;
                if (f) {
                    try_wide = !to_exact && lit_streak >= WIDE_AFTER;
                    continue;
                }
            }
        }
        if (MODE == MODE_ESTIMATE) {
            if (est > peak)
                peak = est; // the reference's loop-top check sees this value (lz_diff.cpp:868-869)
        }
        const uint32_t max_len = n - i;
        uint32_t s0;
        if (have_cands)
            s0 = (uint32_t)wtext[i];
        else {
        // ---- exact step at position i.  key at text[i .. i+key_len)  (get_code, lz_diff.h:58-106) ----
        force_exact = false;
        const uint32_t s = lane < key_len ? (uint32_t)wtext[i + lane] : 0u;
        const uint64_t bad = __ballot(s > 3);
        s0 = bcast_u32(s, 0);

        if (bad) {
            // N-run? (get_Nrun_len, lz_diff.h:122-132)
            uint32_t nrun = 0;
            if ((__ballot(s == N_CODE) & 7ULL) == 7ULL) {
                nrun = max_len; // runs to the end unless a non-N is found
                for (uint32_t base = 3; base < max_len; base += WAVE) {
                    const uint32_t p = base + lane;
                    const bool not_n = p < max_len ? sv_sym(tv, i + p, t_clean) != N_CODE : true;
                    const uint64_t m = __ballot(not_n);
                    if (m) {
                        uint32_t e = base + ctz64(m);
                        nrun = e < max_len ? e : max_len;
                        break;
                    }
                }
            }
            if (nrun >= MIN_NRUN_LEN) {
                if (MODE == MODE_ENCODE) {
                    if (writer)
                        out[o] = N_RUN_STARTER;
                    o += 1;
                    o += emit_uint(out, o, nrun - MIN_NRUN_LEN, writer);
                    if (writer)
                        out[o] = N_CODE;
                    o += 1;
                } else if (MODE == MODE_ESTIMATE) {
                    est += 2 + v2_uint_len(nrun); // lz_diff.h:407-410
                } else {
                    const uint32_t tc = 2 + base_int_len(nrun - MIN_NRUN_LEN);
                    for (uint32_t t = lane; t < nrun; t += WAVE)
                        PUT_COST(o + t, 0);
                    __builtin_amdgcn_s_waitcnt(0); // the run's zeros land before lane 0 stores its cost
                    if (writer)
                        PUT_COST(prefix_costs ? o : o + nrun - 1, (cost_t)tc);
                    o += nrun;
                }
                i += nrun;
                npl = 0;
                try_wide = false;
                lit_streak = 0;
            } else {
                if (MODE == MODE_ENCODE) {
                    if (writer)
                        out[o] = (uint8_t)('A' + s0);
                } else if (MODE == MODE_ESTIMATE)
                    ++est;
                else if (writer)
                    PUT_COST(o, 1);
                ++o;
                ++i;
                ++pred_pos;
                ++npl;
                try_wide = ++lit_streak >= WIDE_AFTER;
            }
            continue;
        }

        // 2-bit code, first symbol most significant
        const uint64_t m0 = __ballot((s & 1u) != 0), m1 = __ballot((s & 2u) != 0);
        const uint64_t r0 = __brevll(m0) >> (64 - key_len), r1 = __brevll(m1) >> (64 - key_len);
        const uint64_t x = spread_bits(r0) | (spread_bits(r1) << 1);
        const uint64_t h = murmur64(x);
        const uint32_t slot = ((uint32_t)h & ht_mask);

        // ---- find_best_match: 64 lanes = 64 probes ----
        bool is_empty, fp_ok;
        if (rd.is_short) {
            const uint32_t e = tab32[(slot + lane) & ht_mask];
            is_empty = e == 0xFFFFFFFFu;
            epos = e >> 16;
            fp_ok = (e & 0xFFFFu) == (uint32_t)(h >> 48);
        } else {
            const uint64_t e = tab64[(slot + lane) & ht_mask];
            is_empty = e == ~0ULL;
            epos = (uint32_t)(e >> 32);
            fp_ok = (uint32_t)e == (uint32_t)(h >> 32);
        }
        const uint64_t em = __ballot(is_empty);
        cand = __ballot(fp_ok && !is_empty);
        if (em)
            cand &= (1ULL << ctz64(em)) - 1ULL; // probes stop at the first empty slot
        }

        PH(3) // literals of the grouped probe / the exact step's probe
        uint32_t len_bck = 0, len_fwd = 0, match_pos = 0;
        uint32_t min_to_update = mml;
        uint32_t best_tb = 0;      // lane b: text symbol at i-1-b for the chosen candidate
        bool best_eq = false;      // lane b: that symbol equals ref[h_pos-1-b]
        while (cand) {
            const uint32_t j = ctz64(cand);
            cand &= cand - 1;
            const uint32_t h_pos = bcast_u32(epos, j) * HASHING_STEP;
            const uint32_t lim = npl < h_pos ? npl : h_pos;
            uint32_t tb = 0x100u, rb = 0x200u; // lanes >= lim: never equal
            uint32_t f_len;
            {
                // backward symbols of the first 64 positions (lz_diff.cpp:308-311), issued before the forward
                // compare so that both arrive in one round trip
                if (lane < lim) {
                    tb = sv_sym(tv, i - 1 - lane, t_clean);
                    rb = sv_sym(rv, h_pos - 1 - lane, r_clean);
                }
                // (the reference pads its copy with key_len symbols no text holds, lz_diff.cpp:48-53: a compare ends at the
                // reference's end at the latest -- here by the bound)
                const uint32_t ref_left = ref_size - h_pos;
                f_len = wave_match_fwd(tv, i, t_clean, rv, h_pos, r_clean, max_len < ref_left ? max_len : ref_left, &win);
            }
            if (f_len >= key_len) {
                const uint64_t mm = __ballot(tb != rb);
                uint32_t b_len;
                if (mm)
                    b_len = ctz64(mm);
                else
                    b_len = lim <= WAVE ? lim : WAVE + wave_common_suffix(tv, i - WAVE, rv, h_pos - WAVE, lim - WAVE);
                if (b_len + f_len > min_to_update) {
                    len_bck = b_len;
                    len_fwd = f_len;
                    match_pos = h_pos;
                    min_to_update = b_len + f_len;
                    best_tb = tb;
                    best_eq = tb == rb;
                }
            }
        }

        PH(4) // candidate verification
        if (len_bck + len_fwd < mml) {
            // literal
            if (MODE == MODE_ENCODE) {
                if (writer)
                    out[o] = (uint8_t)('A' + s0);
            } else if (MODE == MODE_ESTIMATE)
                ++est;
            else if (writer)
                PUT_COST(o, 1);
            ++o;
            ++i;
            ++pred_pos;
            ++npl;
            try_wide = ++lit_streak >= WIDE_AFTER;
            continue;
        }

        try_wide = false;
        lit_streak = 0;
        const uint32_t len = len_bck + len_fwd;
        if (MODE == MODE_ESTIMATE) {
            // no roll-back of the back extension here (lz_diff.cpp:926-936)
            if (i + len == n && match_pos + len == ref_size)
                est += v2_cost_match(mml, match_pos, ~0u, pred_pos);
            else
                est += v2_cost_match(mml, match_pos, len, pred_pos);
            pred_pos = match_pos + len;
            i += len;
            npl = 0;
            LOG_STATE();
            continue;
        }

        // roll the back extension back (lz_diff.cpp:756-766 / 246-256)
        o -= len_bck;
        match_pos -= len_bck;
        pred_pos -= len_bck;
        i -= len_bck;

        if (MODE == MODE_ENCODE) {
            const uint32_t n_trail = npl - len_bck; // literal bytes now ending the delta
            if (coop_out) {
                // stores of other lanes (the wide probe's literal runs) to bytes that are about to be re-written by lane 0
                // (rolled-back or patched literals) must have landed first
                __builtin_amdgcn_s_waitcnt(0);
                coop_out = false;
            }
            stale_out = stale_out || len_bck != 0;
            if (match_pos == pred_pos && n_trail) {
                // Literals equal to the reference become '!' (lz_diff.cpp:769-779).  The reference walks
                // back over the delta while the bytes are 'A'..'Z' (t < e_size, t < match_pos); those
                // bytes are exactly the trailing literals, i.e. text[i-t] -- lane b of the chosen
                // candidate's backward probe holds text[i'-1-b], i' = i before the roll-back, so
                // t = b - len_bck + 1 and nothing has to be read back from the delta.
                const uint32_t b = lane;
                const uint32_t t = b - len_bck + 1;
                const bool in_first = b >= len_bck && b < WAVE && t <= n_trail && t < o && t < match_pos;
                // t <= n_trail implies b <= npl-1 and t < match_pos implies b < h_pos-1: only lanes with real data
                const uint64_t brk = __ballot(in_first && best_tb >= 26);
                const uint32_t first_brk = brk ? ctz64(brk) : WAVE;
                // (lane 0 stores the patches too: it wrote the literals they replace)
                uint64_t pm = __ballot(in_first && b < first_brk && best_eq);
                while (pm) {
                    const uint32_t pb = ctz64(pm);
                    pm &= pm - 1;
                    if (writer)
                        out[o - (pb - len_bck + 1)] = '!';
                }
                // trailing literals beyond the 64 probed ones (rare): serial walk on text/ref
                if (!brk && len_bck + n_trail > WAVE && writer) {
                    for (uint32_t tt = len_bck >= WAVE ? 1u : WAVE - len_bck + 1; tt <= n_trail && tt < o && tt < match_pos; ++tt) {
                        const uint32_t c = sv_sym(tv, i - tt);
                        if (c >= 26)
                            break;
                        if (c == sv_sym(rv, match_pos - tt))
                            out[o - tt] = '!';
                    }
                }
            }
            const bool to_end = (i + len == n) && (match_pos + len == ref_size);
            o += emit_int(out, o, (int32_t)((int)match_pos - (int)pred_pos), writer);
            if (!to_end) {
                if (writer)
                    out[o] = ',';
                o += 1;
                o += emit_uint(out, o, len - mml, writer);
            }
            if (writer)
                out[o] = '.';
            o += 1;
        } else {
            const uint32_t tc = base_cost_match(mml, match_pos, len, pred_pos);
            // lane 0's earlier literal costs (now rolled back) must land before the zero fill,
            // and the zero fill before lane 0's match cost: drain the wave's stores in between
            __builtin_amdgcn_s_waitcnt(0);
            for (uint32_t t = lane; t < len; t += WAVE)
                PUT_COST(o + t, 0);
            __builtin_amdgcn_s_waitcnt(0);
            if (writer)
                PUT_COST(prefix_costs ? o : o + len - 1, (cost_t)tc);
            o += len;
        }
        pred_pos = match_pos + len;
        i += len;
        npl = 0;
        LOG_STATE();
    }
    if (ROLE == 1) { // a chunk ends where its range does: the tail of the text is the hop parser's
        if (writer)
            *cc.log_n = n_log;
        return res;
    }

#ifdef AGC_PHASES
    if (MODE == MODE_ENCODE && writer)
        for (int k = 0; k < 8; ++k) {
            atomicAdd(&g_phase_acc[k], (unsigned long long)ph_acc[k]);
            atomicAdd(&g_phase_cnt[k], (unsigned long long)ph_cnt[k]);
        }
#endif
    // tail literals (lz_diff.cpp:795-796 / 943 / 282-283)
    if (MODE == MODE_ESTIMATE) {
        est += n - i; // u32 wrap-around exactly as the reference
        res.value = est;
        res.peak = peak;
        return res;
    }
    if (i < n) {
        const uint32_t cnt = n - i; // <= key_len symbols (or the whole text when n <= key_len)
        if (MODE == MODE_ENCODE) {
            // one lane owns every byte of the delta: no cross-lane stores to one address
            if (writer)
                for (uint32_t t = 0; t < cnt; ++t)
                    out[o + t] = (uint8_t)('A' + sv_sym(tv, i + t));
        } else {
            __builtin_amdgcn_s_waitcnt(0);
            for (uint32_t t = lane; t < cnt; t += WAVE)
                costs[o + t] = 1;
        }
        o += cnt;
    }
    res.value = o;
    if (ROLE == 2)
        res.peak = (dbg_hops << 16) | (dbg_own < 65535u ? dbg_own : 65535u); // (cost vectors / encode: a debugging aid, AGC_HIP_CHUNK_LOG)
    return res;
#undef PUT_COST
#undef LOG_STATE
}

// One wavefront per segment: wave w of block b parses segment 4*b + w of the host's
// longest-first list (the dispatcher hands blocks out in order, so the long segments start
// first and the short ones fill the tail).
// One wavefront per segment: wave w of block b parses segment 4*b + w of the host's
// longest-first list (the dispatcher hands blocks out in order, so the long segments start
// first and the short ones fill the tail).
template <int MODE>
__global__ void __launch_bounds__(256) lz_parse_kernel(const RefDesc *__restrict__ refs, const SegDesc *__restrict__ segs,
                                                       uint32_t n_segs, uint8_t *__restrict__ out_bytes,
                                                       uint32_t *__restrict__ out_u32, uint32_t *__restrict__ res_value,
                                                       uint32_t *__restrict__ res_peak, const uint32_t *__restrict__ n_segs_dev)
{
    const uint32_t idx = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    AGC_TRACE(1, idx);
    // (n_segs_dev: the number of descriptors is known on the device only -- they were made there, seg_kernels.hip -- and the grid
    // covers an upper bound)
    if (idx >= (n_segs_dev ? *n_segs_dev : n_segs))
        return;
    const SegDesc &sdm = segs[idx];
    const RefDesc &rdm = refs[uniform_u32(sdm.ref_slot)];
    // (everything in the two descriptors is the same for the whole wave: scalar registers)
    RefDesc rd;
    rd.words = uniform_ptr(rdm.words);
    rd.esc_index = uniform_ptr(rdm.esc_index);
    rd.esc_bytes = uniform_ptr(rdm.esc_bytes);
    rd.table = uniform_ptr(rdm.table);
    rd.bloom = uniform_ptr(rdm.bloom);
    rd.ref_size = uniform_u32(rdm.ref_size);
    rd.ht_mask = uniform_u32(rdm.ht_mask);
    rd.key_len = uniform_u32(rdm.key_len);
    rd.min_match_len = uniform_u32(rdm.min_match_len);
    rd.is_short = uniform_u32(rdm.is_short);
    rd.valid = 1;
    const SymViewG tv = global_view(uniform_view(sdm.text));
    const uint64_t out_off = uniform_u64(sdm.out_off);
    const unsigned long long *maybe = uniform_ptr(sdm.maybe);
    const uint32_t flags = uniform_u32(sdm.flags), oidx = uniform_u32(sdm.idx);
    AGC_TRACE(2, tv.len);
    __shared__ __attribute__((aligned(32))) uint8_t s_win[4][WIN_SYMS];
    uint8_t *win_lds = s_win[threadIdx.x >> 6];
    ParseOut r;
    if (MODE == MODE_ENCODE)
        r = lz_parse<MODE>(rd, tv, out_bytes + out_off, nullptr, false, win_lds, nullptr);
    else if (MODE == MODE_ESTIMATE)
        r = lz_parse<MODE>(rd, tv, nullptr, nullptr, false, win_lds, maybe);
    else
        r = lz_parse<MODE>(rd, tv, nullptr, (cost_t *)out_u32 + out_off, (flags & 1u) != 0, win_lds, maybe);
    AGC_TRACE(9, r.value);
    if (lane_id() == 0) {
        res_value[oidx] = r.value;
        if (MODE == MODE_ESTIMATE)
            res_peak[oidx] = r.peak;
    }
}

template __global__ void lz_parse_kernel<MODE_ENCODE>(const RefDesc *, const SegDesc *, uint32_t, uint8_t *, uint32_t *, uint32_t *, uint32_t *, const uint32_t *);
template __global__ void lz_parse_kernel<MODE_ESTIMATE>(const RefDesc *, const SegDesc *, uint32_t, uint8_t *, uint32_t *, uint32_t *, uint32_t *, const uint32_t *);
template __global__ void lz_parse_kernel<MODE_COSTVEC>(const RefDesc *, const SegDesc *, uint32_t, uint8_t *, uint32_t *, uint32_t *, uint32_t *, const uint32_t *);

// ---- the parse in chunks (see ChunkCtl): a wavefront per chunk, then a wavefront per text that hops from chunk to chunk ----
struct ChunkJob {
    uint32_t seg;   // index into the descriptor array
    uint32_t chunk; // chunk of that text
};
struct ChunkPlan {
    const ChunkJob *jobs;        // every chunk of every text (lz_chunk_kernel: one wavefront each)
    const uint32_t *seg_chunk0;  // per descriptor: index of its first chunk in `jobs`, n_segs + 1 entries
    uint32_t n_jobs, chunk_len, cap;
    ChunkState *logs;            // n_jobs x cap
    uint32_t *log_n;             // n_jobs
    uint8_t *chunk_out;          // encode: n_jobs x chunk_stride bytes
    uint32_t chunk_stride;
};

__device__ __forceinline__ void load_descs(const RefDesc *refs, const SegDesc &sdm, RefDesc &rd, SymViewG &tv)
{
    const RefDesc &rdm = refs[uniform_u32(sdm.ref_slot)];
    rd.words = uniform_ptr(rdm.words);
    rd.esc_index = uniform_ptr(rdm.esc_index);
    rd.esc_bytes = uniform_ptr(rdm.esc_bytes);
    rd.table = uniform_ptr(rdm.table);
    rd.bloom = uniform_ptr(rdm.bloom);
    rd.ref_size = uniform_u32(rdm.ref_size);
    rd.ht_mask = uniform_u32(rdm.ht_mask);
    rd.key_len = uniform_u32(rdm.key_len);
    rd.min_match_len = uniform_u32(rdm.min_match_len);
    rd.is_short = uniform_u32(rdm.is_short);
    rd.valid = 1;
    tv = global_view(uniform_view(sdm.text));
}

template <int MODE>
__global__ void __launch_bounds__(256) lz_chunk_kernel(const RefDesc *__restrict__ refs, const SegDesc *__restrict__ segs, ChunkPlan pl,
                                                       uint32_t *__restrict__ out_u32)
{
    const uint32_t j = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (j >= pl.n_jobs)
        return;
    const uint32_t seg = uniform_u32(pl.jobs[j].seg), ch = uniform_u32(pl.jobs[j].chunk);
    const SegDesc &sdm = segs[seg];
    RefDesc rd;
    SymViewG tv;
    load_descs(refs, sdm, rd, tv);
    const uint64_t out_off = uniform_u64(sdm.out_off);
    const unsigned long long *maybe = uniform_ptr(sdm.maybe);
    const uint32_t flags = uniform_u32(sdm.flags);
    __shared__ __attribute__((aligned(32))) uint8_t s_win[4][WIN_SYMS];
    uint8_t *win_lds = s_win[threadIdx.x >> 6];
    ChunkCtl cc;
    cc.i0 = ch * pl.chunk_len;
    cc.i_stop = cc.i0 + pl.chunk_len < tv.len ? cc.i0 + pl.chunk_len : tv.len;
    cc.log = pl.logs + (size_t)j * pl.cap;
    cc.log_n = pl.log_n + j;
    cc.cap = pl.cap;
    cc.chunk_len = pl.chunk_len;
    cc.n_chunks = 0;
    cc.chunk_out = nullptr;
    cc.chunk_stride = 0;
    if (MODE == MODE_ENCODE)
        (void)lz_parse<MODE, 1>(rd, tv, pl.chunk_out + (size_t)j * pl.chunk_stride, nullptr, false, win_lds, nullptr, cc);
    else if (MODE == MODE_ESTIMATE)
        (void)lz_parse<MODE, 1>(rd, tv, nullptr, nullptr, false, win_lds, maybe, cc);
    else
        (void)lz_parse<MODE, 1>(rd, tv, nullptr, (cost_t *)out_u32 + out_off, (flags & 1u) != 0, win_lds, maybe, cc);
}

template <int MODE>
__global__ void __launch_bounds__(256) lz_hop_kernel(const RefDesc *__restrict__ refs, const SegDesc *__restrict__ segs, uint32_t n_segs, ChunkPlan pl,
                                                     uint8_t *__restrict__ out_bytes, uint32_t *__restrict__ out_u32, uint32_t *__restrict__ res_value,
                                                     uint32_t *__restrict__ res_peak)
{
    const uint32_t idx = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (idx >= n_segs)
        return;
    const SegDesc &sdm = segs[idx];
    RefDesc rd;
    SymViewG tv;
    load_descs(refs, sdm, rd, tv);
    const uint64_t out_off = uniform_u64(sdm.out_off);
    const unsigned long long *maybe = uniform_ptr(sdm.maybe);
    const uint32_t flags = uniform_u32(sdm.flags), oidx = uniform_u32(sdm.idx);
    __shared__ __attribute__((aligned(32))) uint8_t s_win[4][WIN_SYMS];
    uint8_t *win_lds = s_win[threadIdx.x >> 6];
    const uint32_t j0 = uniform_u32(pl.seg_chunk0[idx]), j1 = uniform_u32(pl.seg_chunk0[idx + 1]);
    ChunkCtl cc;
    cc.i0 = 0;
    cc.i_stop = 0xFFFFFFFFu;
    cc.log = pl.logs + (size_t)j0 * pl.cap;
    cc.log_n = pl.log_n + j0;
    cc.cap = pl.cap;
    cc.chunk_len = pl.chunk_len;
    cc.n_chunks = j1 - j0;
    cc.chunk_out = pl.chunk_out + (size_t)j0 * pl.chunk_stride;
    cc.chunk_stride = pl.chunk_stride;
    ParseOut r;
    if (MODE == MODE_ENCODE)
        r = lz_parse<MODE, 2>(rd, tv, out_bytes + out_off, nullptr, false, win_lds, nullptr, cc);
    else if (MODE == MODE_ESTIMATE)
        r = lz_parse<MODE, 2>(rd, tv, nullptr, nullptr, false, win_lds, maybe, cc);
    else
        r = lz_parse<MODE, 2>(rd, tv, nullptr, (cost_t *)out_u32 + out_off, (flags & 1u) != 0, win_lds, maybe, cc);
    if (lane_id() == 0) {
        res_value[oidx] = r.value;
        res_peak[oidx] = r.peak;
    }
}

template __global__ void lz_chunk_kernel<MODE_ENCODE>(const RefDesc *, const SegDesc *, ChunkPlan, uint32_t *);
template __global__ void lz_chunk_kernel<MODE_ESTIMATE>(const RefDesc *, const SegDesc *, ChunkPlan, uint32_t *);
template __global__ void lz_chunk_kernel<MODE_COSTVEC>(const RefDesc *, const SegDesc *, ChunkPlan, uint32_t *);
template __global__ void lz_hop_kernel<MODE_ENCODE>(const RefDesc *, const SegDesc *, uint32_t, ChunkPlan, uint8_t *, uint32_t *, uint32_t *, uint32_t *);
template __global__ void lz_hop_kernel<MODE_ESTIMATE>(const RefDesc *, const SegDesc *, uint32_t, ChunkPlan, uint8_t *, uint32_t *, uint32_t *, uint32_t *);
template __global__ void lz_hop_kernel<MODE_COSTVEC>(const RefDesc *, const SegDesc *, uint32_t, ChunkPlan, uint8_t *, uint32_t *, uint32_t *, uint32_t *);

// gathers the per-segment deltas (scratch slots) into one contiguous buffer
__global__ void __launch_bounds__(256) gather_bytes_kernel(const uint8_t *__restrict__ scratch, const SegDesc *__restrict__ segs,
                                                           const uint32_t *__restrict__ lens, const uint64_t *__restrict__ dst_off,
                                                           uint32_t n_segs, uint8_t *__restrict__ dst)
{
    for (uint32_t s = blockIdx.x; s < n_segs; s += gridDim.x) {
        const uint32_t orig = segs[s].idx;
        const uint8_t *src = scratch + segs[s].out_off;
        uint8_t *d = dst + dst_off[orig];
        const uint32_t n = lens[orig];
        for (uint32_t t = threadIdx.x; t < n; t += blockDim.x)
            d[t] = src[t];
    }
}

// ---------------------------------------------------------------------------
// Key filter: for every position of a text, may the key starting there be in the reference's index?
// One block per (text, chunk of FILTER_CHUNK positions); the reference's filter (32 KiB) sits in LDS.  Each lane takes 16
// consecutive positions per step: one 16-byte load, 2-bit packing, the two following lanes' words by shuffles (the last two
// lanes of a wave load their followers themselves), then per position: key = a funnel shift of the packed window, three
// multiplies for the filter hash, one LDS read.  Bit = 1 ("may match / let the exact step decide") when the key is in the
// filter, contains a symbol outside ACGT, or runs past the end of the text.
// ---------------------------------------------------------------------------
constexpr uint32_t FILTER_CHUNK = 65536;

struct FilterJob {
    SymView text;
    const unsigned long long *bloom;
    unsigned long long *out;   // (len + 63) / 64 words
    uint32_t key_len;
    uint32_t chunk;            // first position of this block's chunk
    uint32_t bloom_shift;      // key_bloom_shift of the reference
    uint32_t pad;
};

template <class V> __device__ __forceinline__ void pack16(const V &tv, uint32_t pos, uint32_t &P, uint32_t &I)
{
    // 16 symbols at pos: P = 2-bit codes (first symbol most significant), I = mask of symbols > 3 (first symbol = bit 15);
    // positions at or after len count as invalid
    P = 0;
    I = 0xFFFF;
    if (pos < tv.len) {
        const uint32_t cnt = tv.len - pos < 16 ? tv.len - pos : 16;
        uint64_t Q;
        uint32_t J;
        sv_fetch32(tv, pos, cnt, false, Q, J);
        if (cnt < 16) {
            J |= 0xFFFFu << cnt;
            Q &= (1ULL << (2 * cnt)) - 1ULL;
        }
        P = (uint32_t)(sv_rev2_64(Q) >> 32);
        I = sv_brev32(J & 0xFFFFu) >> 16;
    }
}

// One pass of the first filter over the 16 positions of every lane (bit j of the result: the key at position j is in it).
// V2:V1:V0 = the lane's 96-bit window shifted so that the key of position 15 starts at bit 0; the key of position j is then a
// funnel shift by the constant 2 (15 - j).
template <bool IN_LDS>
__device__ __forceinline__ uint32_t key_filter_pass1(uint32_t V0, uint32_t V1, uint32_t V2, uint32_t kmask_lo, uint32_t kmask_hi, uint32_t bshift,
                                                     const uint32_t *s_filter, g_u32 *g_filter)
{
    uint32_t pass = 0;
#pragma unroll
    for (int h = 1; h >= 0; --h) { // (eight reads in flight before the first outcome is needed)
        uint32_t fw[8], bm[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int sh = 2 * (15 - (8 * h + i));
            const uint32_t lo = (sh ? __builtin_amdgcn_alignbit(V1, V0, sh) : V0) & kmask_lo;
            const uint32_t hi = (sh ? __builtin_amdgcn_alignbit(V2, V1, sh) : V1) & kmask_hi;
            uint32_t bw;
            key_bloom_slot(((uint64_t)hi << 32) | lo, bshift, bw, bm[i]);
            fw[i] = IN_LDS ? s_filter[bw] : g_filter[bw];
        }
#pragma unroll
        for (int i = 7; i >= 0; --i) {
            const uint32_t missing = bm[i] & ~fw[i];
            asm("v_cmp_eq_u32_e32 vcc, 0, %1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(pass) : "v"(missing) : "vcc");
        }
    }
    return pass;
}

// (256 threads per copy of the filter: 512 / 1024 were measured -- half / a quarter of the LDS per wavefront beside the scan's 128 KiB,
// the kernel alone twice as fast -- but a block that needs 8 / 16 free wave slots on ONE CU at once waits for them while the
// whole-sample encode holds the slots: the step's filter row 3.0-3.5 ms instead of 2.7, profiles/EXPERIMENTS.md)
constexpr uint32_t FILTER_THREADS = 256;

__global__ void __launch_bounds__(FILTER_THREADS) key_filter_kernel(const FilterJob *__restrict__ jobs)
{
    const FilterJob jb = jobs[blockIdx.x];
    const SymViewG text = global_view(jb.text); // (global loads instead of FLAT ones: see SymViewG)
    g_u32 *bloom = (g_u32 *)jb.bloom;
    unsigned long long __attribute__((address_space(1))) *out = (unsigned long long __attribute__((address_space(1))) *)jb.out;
    // (the first filter in LDS; the second one is consulted for the keys that pass it -- a quarter of the positions of a text that
    // matches its reference, 1 % of a foreign one's -- from HBM / L2)
    // (a filter of more than KEY_BLOOM_HALF words -- a reference of more than 64 k symbols -- is read where it lies: L2)
    __shared__ __attribute__((aligned(16))) uint32_t s_bloom[2 * KEY_BLOOM_HALF];
    const uint32_t bshift = jb.bloom_shift;
    const bool in_lds = bshift == KEY_BLOOM_SHIFT0;
    if (in_lds)
        for (uint32_t t = threadIdx.x; t < 2 * KEY_BLOOM_HALF; t += blockDim.x)
            s_bloom[t] = bloom[t];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t k = jb.key_len;                        // key_len <= 29
    const uint64_t kmask = (1ULL << (2 * k)) - 1ULL;
    const uint32_t kmask_lo = (uint32_t)kmask, kmask_hi = (uint32_t)(kmask >> 32);
    const uint32_t imask = (1u << k) - 1u;
    const uint32_t r0 = 66 - 2 * k;                       // 8 .. 58: the key of position 15 starts r0 bits above the window's end
    const uint32_t len = text.len;
    const uint32_t end = min(len, jb.chunk + FILTER_CHUNK);
    for (uint32_t base = jb.chunk + wave * 1024; base < end; base += (blockDim.x >> 6) * 1024) {
        const uint32_t pos = base + lane * 16;
        uint32_t P, I;
        pack16(text, pos, P, I);
        uint32_t P1 = __shfl_down(P, 1), I1 = __shfl_down(I, 1), P2 = __shfl_down(P, 2), I2 = __shfl_down(I, 2);
        if (lane >= 62) { // the followers of the last two lanes belong to the next step
            if (lane == 63)
                pack16(text, pos + 16, P1, I1);
            pack16(text, pos + 32, P2, I2);
        }
        // 96-bit window P:P1:P2: symbols 0..15 (P), 16..31 (P1), 32..47 (P2); symbol s at bits [94 - 2s, 95 - 2s]
        uint32_t V0, V1, V2;
        if (r0 < 32) {
            V0 = __builtin_amdgcn_alignbit(P1, P2, r0);
            V1 = __builtin_amdgcn_alignbit(P, P1, r0);
            V2 = P >> r0;
        } else {
            V0 = __builtin_amdgcn_alignbit(P, P1, r0 - 32);
            V1 = P >> (r0 - 32);
            V2 = 0;
        }
        uint32_t cand = in_lds ? key_filter_pass1<true>(V0, V1, V2, kmask_lo, kmask_hi, bshift, s_bloom, bloom)
                               : key_filter_pass1<false>(V0, V1, V2, kmask_lo, kmask_hi, bshift, s_bloom, bloom);
        // keys that hold a symbol outside ACGT or run past the end of the text: "let the exact step decide" (the text's last lanes,
        // N runs)
        uint32_t bits = 0;
        const uint64_t inv = ((uint64_t)I << 32) | ((uint64_t)I1 << 16) | I2; // symbol s <-> bit 47 - s
        if (inv != 0 || pos + 15 + k >= len) {
#pragma unroll
            for (uint32_t j = 0; j < 16; ++j) {
                const bool bad = ((uint32_t)(inv >> (48 - j - k)) & imask) != 0;
                const bool past = !(pos + j + k < len);
                bits |= (uint32_t)(bad || past) << j;
            }
        }
        cand &= ~bits;
        // the second filter, for the keys the first one lets through
        while (cand) {
            const uint32_t j = (uint32_t)__builtin_ctz(cand);
            cand &= cand - 1;
            const uint32_t sh = 2 * (15 - j);
            const uint64_t V10 = ((uint64_t)V1 << 32) | V0, V21 = ((uint64_t)V2 << 32) | V1;
            const uint32_t lo = (uint32_t)(V10 >> sh) & kmask_lo, hi = (uint32_t)(V21 >> sh) & kmask_hi;
            uint32_t bw, bm;
            key_bloom_slot2(((uint64_t)hi << 32) | lo, bshift, bw, bm);
            if ((bloom[bw] & bm) == bm)
                bits |= 1u << j;
        }
        // 4 lanes make one 64-bit word (positions ascending = bits ascending)
        const uint64_t mine = (uint64_t)bits << (16 * (lane & 3));
        uint64_t word = mine | __shfl_xor(mine, 1);
        word |= __shfl_xor(word, 2);
        if ((lane & 3) == 0 && pos < len)
            out[pos >> 6] = word;
    }
}

// ---------------------------------------------------------------------------
// Index build.
// ---------------------------------------------------------------------------
struct IdxBuild {
    SymView src;         // the new reference where it lies in the sample (read reverse-complemented when src.rc)
    uint32_t *words;     // its stored form: 2-bit words from bit 0 (ref_pack_kernel writes them)
    int32_t *esc_index;  // nullptr unless the reference holds a symbol outside ACGT (ref_esc_kernel fills both)
    uint8_t *esc_bytes;
    void *table;         // filled by the insert kernel (pre-set to all ones)
    unsigned long long *bloom; // key filter (zeroed), filled by the insert kernel
    uint32_t ref_size;
    uint32_t key_len;
    uint32_t ht_mask;
    uint32_t is_short;
};

// key at positions [pos, pos + key_len) of a sequence or ~0 when one of its symbols is not ACGT or it runs past the end
// (get_code, lz_diff.h:58-106; the reference's copy ends in key_len symbols no key may hold, lz_diff.cpp:48-53)
__device__ __forceinline__ uint64_t key_at(const SymView &v, uint32_t pos, uint32_t key_len)
{
    if ((uint64_t)pos + key_len > v.len)
        return ~0ULL;
    uint64_t P;
    uint32_t I;
    sv_fetch32(v, pos, key_len, false, P, I);
    if (I)
        return ~0ULL;
    return sv_key_from_packed(P, key_len);
}

// The new references into their stored form: 16 symbols per thread and word, read through the sample's view (any offset,
// either orientation), written from bit 0 of the reference's own words.  flags[job] != 0 afterwards: the reference holds a
// symbol outside ACGT (its escape blocks are then filled by ref_esc_kernel).  The tail words (REF_TAIL_WORDS) are zeroed.
__global__ void __launch_bounds__(256) ref_pack_kernel(const IdxBuild *__restrict__ jobs, uint32_t *__restrict__ flags, uint32_t split)
{
    const uint32_t job = blockIdx.x / split, part = blockIdx.x % split;
    const IdxBuild jb = jobs[job];
    const uint32_t n_words = (jb.ref_size + 15) / 16;
    bool high = false;
    for (uint32_t w = part * blockDim.x + threadIdx.x; w < n_words + REF_TAIL_WORDS; w += blockDim.x * split) {
        uint32_t out = 0;
        if (w < n_words) {
            const uint32_t pos = w * 16, cnt = jb.ref_size - pos < 16 ? jb.ref_size - pos : 16;
            uint64_t P;
            uint32_t I;
            sv_fetch32(jb.src, pos, cnt, false, P, I);
            out = cnt < 16 ? (uint32_t)P & ((1u << (2 * cnt)) - 1u) : (uint32_t)P;
            high = high || I != 0;
        }
        jb.words[w] = out;
    }
    if (high)
        atomicOr(&flags[job], 1u);
}

// escape blocks of a reference that holds symbols outside ACGT: one thread block per (reference, 1024-symbol block)
struct EscJob {
    SymView src;
    int32_t *esc_index;
    uint8_t *esc_bytes;
    uint32_t block;
    uint32_t pad;
};
__global__ void __launch_bounds__(256) ref_esc_kernel(const EscJob *__restrict__ jobs)
{
    const EscJob jb = jobs[blockIdx.x];
    uint8_t c[4];
    bool high = false;
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
        const uint32_t pos = jb.block * SV_BLOCK + threadIdx.x * 4 + j;
        c[j] = pos < jb.src.len ? (uint8_t)sv_sym(jb.src, pos) : (uint8_t)0;
        high = high || c[j] > 3;
    }
    const int any = __syncthreads_or(high ? 1 : 0);
    if (threadIdx.x == 0)
        jb.esc_index[jb.block] = any ? (int32_t)jb.block : -1;
    if (any) {
        uint32_t v;
        __builtin_memcpy(&v, c, 4);
        *(uint32_t *)(jb.esc_bytes + (size_t)jb.block * SV_BLOCK + threadIdx.x * 4) = v;
    }
}

// number of valid keys at positions 0,4,8,... (the count prepare_index sizes the table by,
// lz_diff.cpp:88-101: a key is counted where the last key_len symbols are all ACGT and the
// key starts at a multiple of hashing_step); read from the sample's view (the stored form may still be in the making)
// A reference may be spread over `split` blocks (few, long references): counts[] must be zeroed.
__global__ void __launch_bounds__(256) idx_count_kernel(const IdxBuild *__restrict__ jobs, uint32_t *__restrict__ counts, uint32_t split)
{
    const uint32_t job = blockIdx.x / split, part = blockIdx.x % split;
    const IdxBuild jb = jobs[job];
    uint32_t c = 0;
    for (uint32_t t = part * blockDim.x + threadIdx.x; (uint64_t)t * HASHING_STEP + jb.key_len <= jb.ref_size; t += blockDim.x * split)
        c += key_at(jb.src, t * HASHING_STEP, jb.key_len) != ~0ULL;
    // block reduce
    for (int o = 32; o > 0; o >>= 1)
        c += __shfl_down(c, o);
    __shared__ uint32_t part_sum[4];
    if ((threadIdx.x & 63) == 0)
        part_sum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicAdd(&counts[job], part_sum[0] + part_sum[1] + part_sum[2] + part_sum[3]);
}

// Deterministic parallel linear-probing insertion.  The reference inserts keys in
// increasing position order, each into the first empty slot of its <= 64 probes
// (lz_diff.cpp:375-428).  Equivalent fixpoint: every slot ends up holding the SMALLEST
// position whose probe sequence reaches it without finding an earlier free slot -- so a
// thread claims slots with atomicMin on (pos<<bits | fp) and carries any larger entry it
// displaces onward.  Entries only ever decrease, which makes the result independent of
// thread timing and identical to the sequential build.
template <typename E, int FPBITS>
__device__ void idx_insert_one(const IdxBuild &jb, const SymView &rv, uint32_t t)
{
    const uint64_t x = key_at(rv, t * HASHING_STEP, jb.key_len);
    if (x == ~0ULL)
        return;
    E *tab = (E *)jb.table;
    const uint64_t h = murmur64(x);
    if (jb.bloom) {
        uint32_t bw, bm;
        const uint32_t bshift = key_bloom_shift(jb.ref_size);
        key_bloom_slot(x, bshift, bw, bm);
        atomicOr((uint32_t *)jb.bloom + bw, bm);
        key_bloom_slot2(x, bshift, bw, bm);
        atomicOr((uint32_t *)jb.bloom + bw, bm);
    }
    const E fp = FPBITS == 16 ? (E)(h >> 48) : (E)(h >> 32);
    E cur = ((E)t << FPBITS) | fp;
    uint32_t slot = (uint32_t)h & jb.ht_mask;
    uint32_t tries = 0;
    const E EMPTY = (E)~(E)0;
    for (;;) {
        const E old = atomicMin(&tab[slot], cur);
        ++tries;
        if (old == EMPTY)
            return;
        if (old > cur) {
            // we took the slot from `old`: continue inserting `old` after this slot
            cur = old;
            const uint32_t ot = (uint32_t)(old >> FPBITS);
            const uint64_t ox = key_at(rv, ot * HASHING_STEP, jb.key_len);
            const uint32_t ohome = (uint32_t)murmur64(ox) & jb.ht_mask;
            tries = ((slot - ohome) & jb.ht_mask) + 1;
        }
        if (tries >= MAX_NO_TRIES)
            return; // dropped, as in the reference
        slot = (slot + 1) & jb.ht_mask;
    }
}

__global__ void __launch_bounds__(256) idx_insert_kernel(const IdxBuild *__restrict__ jobs, uint32_t split)
{
    // the atomicMin fixpoint does not depend on which block inserts which key
    const IdxBuild jb = jobs[blockIdx.x / split];
    const SymView rv = {jb.words, jb.esc_index, jb.esc_bytes, 0, jb.ref_size, 0}; // the stored form
    for (uint32_t t = (blockIdx.x % split) * blockDim.x + threadIdx.x; (uint64_t)t * HASHING_STEP < jb.ref_size; t += blockDim.x * split) {
        if (jb.is_short)
            idx_insert_one<uint32_t, 16>(jb, rv, t);
        else
            idx_insert_one<unsigned long long, 32>(jb, rv, t);
    }
}

// ---------------------------------------------------------------------------
// sequences out of a packed buffer as bytes, optionally reverse-complemented (reverse_complement_copy,
// src/common/agc_basic.cpp:282-315): what the host packs itself (new references, raw segments)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) slice_expand_kernel(const ViewJob *__restrict__ jobs, uint32_t n_jobs)
{
    for (uint32_t s = blockIdx.x; s < n_jobs; s += gridDim.x) {
        const ViewJob jb = jobs[s];
        uint8_t *dst = (uint8_t *)jb.dst;
        const uint32_t len = jb.src.len;
        for (uint32_t pos = threadIdx.x * 16; pos < len; pos += blockDim.x * 16) {
            const uint32_t cnt = len - pos < 16 ? len - pos : 16;
            uint64_t P;
            uint32_t I;
            sv_fetch32(jb.src, pos, cnt, false, P, I);
            if (cnt == 16 && I == 0) {
                const uint32_t lo = (uint32_t)P;
                const uint4 v = make_uint4(sv_expand4(lo), sv_expand4(lo >> 8), sv_expand4(lo >> 16), sv_expand4(lo >> 24));
                __builtin_memcpy(dst + pos, &v, 16);
            } else
                for (uint32_t j = 0; j < cnt; ++j)
                    dst[pos + j] = (uint8_t)(((I >> j) & 1u) ? sv_sym(jb.src, pos + j) : (uint32_t)(P >> (2u * j)) & 3u);
        }
    }
}

// repetitiveness probe counters (segment.h:224-247): for lag 4..31,
// cnt = #{j : j+lag < n, d[j]==d[j+lag]}, cur = #{j : j+lag < n, d[j] < 4}
// cnt_out / cur_out must be zeroed; a sequence may be spread over `split` blocks.  A thread takes 32 positions: on clean
// sequence the 28 comparisons are XORs of the 64-bit chunk with itself shifted (the next chunk funnelled in) + a popcount.
__global__ void __launch_bounds__(256) lag_counts_kernel(const ViewJob *__restrict__ jobs, uint32_t *__restrict__ cnt_out,
                                                         uint32_t *__restrict__ cur_out, uint32_t split)
{
    const uint32_t job = blockIdx.x / split, part = blockIdx.x % split;
    const SymView sv = jobs[job].src;
    __shared__ uint32_t s_cnt[28], s_cur[28];
    if (threadIdx.x < 28) {
        s_cnt[threadIdx.x] = 0;
        s_cur[threadIdx.x] = 0;
    }
    __syncthreads();
    uint32_t cnt[28], cur[28];
#pragma unroll
    for (int l = 0; l < 28; ++l)
        cnt[l] = cur[l] = 0;
    const uint32_t n = sv.len;
    for (uint32_t j0 = (part * blockDim.x + threadIdx.x) * 32; j0 < n; j0 += blockDim.x * split * 32) {
        const uint32_t c0 = n - j0 < 32 ? n - j0 : 32;
        const uint32_t c1 = n - j0 > 32 ? (n - j0 - 32 < 32 ? n - j0 - 32 : 32) : 0;
        uint64_t X, Y = 0;
        uint32_t IX, IY = 0;
        sv_fetch32(sv, j0, c0, false, X, IX);
        if (c1)
            sv_fetch32(sv, j0 + 32, c1, false, Y, IY);
        if ((IX | IY) == 0) {
#pragma unroll
            for (int l = 0; l < 28; ++l) {
                const uint32_t lag = 4 + l;
                // positions j0 + t (t < 32) with j0 + t + lag < n
                if (j0 + lag >= n)
                    continue;
                const uint32_t m = n - lag - j0 < 32 ? n - lag - j0 : 32;
                const uint64_t Z = (X >> (2 * lag)) | (Y << (64 - 2 * lag));
                const uint64_t e = ~(X ^ Z);
                uint64_t eq = e & (e >> 1) & 0x5555555555555555ULL;
                if (m < 32)
                    eq &= (1ULL << (2 * m)) - 1ULL;
                cnt[l] += (uint32_t)__builtin_popcountll(eq);
                cur[l] += m;
            }
        } else {
            // symbols outside ACGT around: by value (equal codes count, whatever they are)
            uint8_t s[64];
            for (uint32_t t = 0; t < 64; ++t)
                s[t] = j0 + t < n ? (uint8_t)sv_sym(sv, j0 + t) : (uint8_t)0xFF;
            for (uint32_t t = 0; t < c0; ++t) {
                const uint32_t v = s[t] < 4;
#pragma unroll
                for (int l = 0; l < 28; ++l) {
                    const uint32_t lag = 4 + l;
                    if (j0 + t + lag < n) {
                        cnt[l] += s[t] == s[t + lag];
                        cur[l] += v;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int l = 0; l < 28; ++l) {
        uint32_t a = cnt[l], b = cur[l];
        for (int o = 32; o > 0; o >>= 1) {
            a += __shfl_down(a, o);
            b += __shfl_down(b, o);
        }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&s_cnt[l], a);
            atomicAdd(&s_cur[l], b);
        }
    }
    __syncthreads();
    if (threadIdx.x < 28) {
        atomicAdd(&cnt_out[job * 28 + threadIdx.x], s_cnt[threadIdx.x]);
        atomicAdd(&cur_out[job * 28 + threadIdx.x], s_cur[threadIdx.x]);
    }
}

// ---------------------------------------------------------------------------
// Missing-middle-splitter split point (find_cand_segment_with_missing_middle_splitter,
// src/core/agc_compressor.cpp:1540-1625): given the two cost vectors of one segment,
//   v1[i] = sum_{j<=i} c1[j]           (c1 read reversed when rev1)
//   v2[i] = sum_{j>=i} c2[j]           (c2 read reversed when rev2)
// return the first i minimising v1[i] + v2[i] (u32 arithmetic as std::partial_sum on
// vector<uint32_t>).  One block per segment.
// ---------------------------------------------------------------------------
struct SplitJob {
    uint64_t off1, off2;   // u32 offsets of the two cost vectors in the scratch buffer
    uint32_t n;
    uint32_t rev1, rev2;
    uint32_t pad;
};

// eight consecutive costs of a vector read forwards (c[i0 .. i0+8)) or backwards (c[n-1-i0], c[n-2-i0] ...) in one 8-byte load;
// positions >= n give 0
__device__ inline void split_load8(const cost_t *__restrict__ c, uint32_t n, uint32_t i0, bool rev, uint32_t x[8])
{
    uint64_t w = 0;
    if (i0 + 8 <= n) {
        const cost_t *p = rev ? c + (n - 8 - i0) : c + i0;
        uint64_t v;
        memcpy(&v, p, 8);
        w = rev ? __builtin_bswap64(v) : v;
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j)
            x[j] = (uint32_t)(w >> (8 * j)) & 0xFF;
    } else {
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) {
            const uint32_t i = i0 + j;
            x[j] = i < n ? (uint32_t)(rev ? c[n - 1 - i] : c[i]) : 0u;
        }
    }
}

__global__ void __launch_bounds__(256) split_point_kernel(const SplitJob *__restrict__ jobs, const cost_t *__restrict__ costs,
                                                          uint32_t *__restrict__ best_pos, uint32_t *__restrict__ best_sum)
{
    const SplitJob jb = jobs[blockIdx.x];
    const uint32_t n = jb.n;
    const cost_t *c1 = costs + jb.off1, *c2 = costs + jb.off2;
    __shared__ uint32_t s_a[256], s_b[256];
    __shared__ uint32_t s_carry1, s_carry2, s_total2;
    const uint32_t tid = threadIdx.x;
    constexpr uint32_t PER = 8;

    // total of c2
    uint32_t t2 = 0;
    for (uint32_t base = 0; base < n; base += 256 * PER) {
        uint32_t x[PER];
        split_load8(c2, n, base + tid * PER, false, x);
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j)
            t2 += x[j];
    }
    s_a[tid] = t2;
    __syncthreads();
    for (uint32_t o = 128; o > 0; o >>= 1) {
        if (tid < o)
            s_a[tid] += s_a[tid + o];
        __syncthreads();
    }
    if (tid == 0) {
        s_total2 = s_a[0];
        s_carry1 = 0;
        s_carry2 = 0;
    }
    __syncthreads();
    const uint32_t total2 = s_total2;

    uint32_t my_best = 0xFFFFFFFFu, my_pos = 0xFFFFFFFFu;
    for (uint32_t base = 0; base < n; base += 256 * PER) {
        const uint32_t b = base + tid * PER;
        uint32_t a1[PER], a2[PER];
        uint32_t sum1 = 0, sum2 = 0;
        split_load8(c1, n, b, jb.rev1 != 0, a1);
        split_load8(c2, n, b, jb.rev2 != 0, a2);
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) {
            sum1 += a1[j];
            sum2 += a2[j];
        }
        s_a[tid] = sum1;
        s_b[tid] = sum2;
        __syncthreads();
        // inclusive Hillis-Steele scan over the 256 thread sums
        for (uint32_t o = 1; o < 256; o <<= 1) {
            uint32_t va = 0, vb = 0;
            if (tid >= o) {
                va = s_a[tid - o];
                vb = s_b[tid - o];
            }
            __syncthreads();
            s_a[tid] += va;
            s_b[tid] += vb;
            __syncthreads();
        }
        uint32_t p1 = s_carry1 + s_a[tid] - sum1; // exclusive prefix of c1 before this thread's chunk
        uint32_t p2 = s_carry2 + s_b[tid] - sum2; // exclusive prefix of c2
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) {
            const uint32_t i = b + j;
            p1 += a1[j];                          // inclusive prefix of c1 at i
            const uint32_t v2 = total2 - p2;      // suffix sum of c2 at i
            p2 += a2[j];
            if (i < n) {
                const uint32_t cs = p1 + v2;
                if (cs < my_best) {
                    my_best = cs;
                    my_pos = i;
                }
            }
        }
        __syncthreads();
        if (tid == 255) {
            s_carry1 += s_a[255];
            s_carry2 += s_b[255];
        }
        __syncthreads();
    }
    // arg-min, first position on ties
    s_a[tid] = my_best;
    s_b[tid] = my_pos;
    __syncthreads();
    for (uint32_t o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            const uint32_t ob = s_a[tid + o], op = s_b[tid + o];
            if (ob < s_a[tid] || (ob == s_a[tid] && op < s_b[tid])) {
                s_a[tid] = ob;
                s_b[tid] = op;
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        best_pos[blockIdx.x] = n ? s_b[0] : 0;
        best_sum[blockIdx.x] = s_a[0];
    }
}

} // namespace agc
