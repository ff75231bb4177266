// lz_group.hip -- the LZ-diff ENCODE parse with 16 lanes per segment (included by api.hip after lz_kernels.hip).
//
// lz_parse_kernel (lz_kernels.hip) gives a whole wavefront to one segment.  In the match / SNP / match rhythm of a sample that
// differs from its reference once per ~1000 symbols that wave executes ~860 instructions per edit, nearly all of them with one
// useful lane (hash of a key, a probe of four table rows, decimal digits of a token, scalar bookkeeping): the encode is bound by
// instruction issue (VALU 51 % + SALU 54 % busy), not by HBM.  Here a wavefront parses FOUR segments, one per group
// of 16 lanes, all of the parse's state in vector registers: the same instruction stream serves four edits.
//
// The group parse handles the plain case only -- text and reference without any symbol outside ACGT (no escaped block, so no
// N runs and no invalid keys: get_code / get_Nrun_len, lz_diff.h:58-132, reduce to the 2-bit key) -- and hands everything
// else to the wave kernel through a deferred list.  Its result is the same delta, byte for byte (CLZDiff_V2::Encode,
// src/common/lz_diff.cpp:655-798; find_best_match, lz_diff.cpp:286-347).
//
// Text: a packed window of 2048 symbols per group in LDS, ORIENTED (a reverse-complement view is turned while the window is
// filled: ~rev2 of the buffer's words in descending order), so that keys, compares in both directions and literals are plain
// funnel shifts of LDS words.  Reference: packed words from HBM, 32 symbols (three dwords) per lane and compare step.
namespace agc {

constexpr uint32_t GRP = 16;
constexpr uint32_t GRP_STEP = GRP * 32; // symbols one forward-compare step of a group covers
constexpr uint32_t GW_WORDS = 128;      // window: 2048 symbols
constexpr uint32_t GW_SYMS = GW_WORDS * 16;
constexpr uint32_t GW_BACK = 64;        // symbols kept before the position a refill is asked for (back extension)

// (pointers with their address space spelled out: through a descriptor loaded from memory the compiler sees generic pointers and
// emits FLAT loads -- for the LDS window that is a trip through the vector memory pipeline per key instead of a ds_read)
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef const __attribute__((address_space(1))) uint32_t glb_u32;
typedef const __attribute__((address_space(1))) uint64_t glb_u64;

__device__ __forceinline__ uint64_t grp_raw64(glb_u32 *words, uint64_t s) // sv_raw64 on a global pointer
{
    glb_u32 *p = words + (s >> 4);
    const uint32_t sh = 2u * (uint32_t)(s & 15u);
    const uint32_t a = p[0], b = p[1], c = p[2];
    uint64_t x = (((uint64_t)b << 32) | a) >> sh;
    if (sh)
        x |= (uint64_t)c << (64 - sh);
    return x;
}

struct GrpText {
    SymView tv;     // the text (the same in the 16 lanes)
    lds_u32 *lds;   // GW_WORDS + 4 words of LDS owned by the group
    int32_t base;   // text position of field 0 of window word 0 (negative: the word starts before the text)
    bool filled;
};

__device__ __forceinline__ uint32_t grp_ballot(bool p) { return (uint32_t)(__ballot(p) >> (lane_id() & 48u)) & 0xFFFFu; }
__device__ __forceinline__ uint32_t grp_bcast(uint32_t v, uint32_t j) { return (uint32_t)__shfl((int)v, (int)((lane_id() & 48u) + j)); }

__device__ __forceinline__ bool grp_has(const GrpText &g, uint32_t p, uint32_t cnt)
{
    return g.filled && (int32_t)p >= g.base && (int32_t)(p + cnt) <= g.base + (int32_t)GW_SYMS;
}

// window <- text positions from (pos - GW_BACK) on, aligned to a word of the buffer; eight words per lane
__device__ void grp_fill(GrpText &g, uint32_t pos)
{
    const uint32_t gl = lane_id() & 15u;
    const SymView &v = g.tv;
    const uint32_t p0 = pos < GW_BACK ? 0u : pos - GW_BACK;
    const uint64_t wmin = v.start >> 4, wmax = (v.start + v.len - 1u) >> 4; // the buffer words that hold the text (len >= 1)
    glb_u32 *tw = (glb_u32 *)v.words;
    uint32_t w[8];
    if (!v.rc) {
        const uint64_t s_al = (v.start + p0) & ~15ULL;
        g.base = (int32_t)((int64_t)s_al - (int64_t)v.start);
        const uint64_t w0 = (s_al >> 4) + 8u * gl;
        if (w0 + 7 <= wmax) {
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k)
                w[k] = tw[w0 + k];
        } else {
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k)
                w[k] = w0 + k <= wmax ? tw[w0 + k] : 0u;
        }
    } else {
        // position q <-> buffer index start + len - 1 - q: window word t is buffer word E/16 - 1 - t with its fields reversed and
        // complemented, E = the buffer index one past the window's first position, rounded up to a word
        const uint64_t E = (v.start + v.len - p0 + 15u) & ~15ULL;
        g.base = (int32_t)((int64_t)(v.start + v.len) - (int64_t)E);
        const int64_t wt = (int64_t)(E >> 4) - 1 - 8 * (int64_t)gl;
        if (wt - 7 >= (int64_t)wmin) {
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k)
                w[k] = tw[wt - (int64_t)k];
        } else {
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k)
                w[k] = wt - (int64_t)k >= (int64_t)wmin ? tw[wt - (int64_t)k] : 0u;
        }
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k)
            w[k] = ~sv_rev2_32(w[k]);
    }
#pragma unroll
    for (uint32_t k = 0; k < 8; ++k)
        g.lds[8u * gl + k] = w[k];
    g.filled = true;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // (one wave: the LDS queue keeps the order; this keeps the compiler's)
    __builtin_amdgcn_wave_barrier();
}

// 32 symbols from text position p, which the window holds (grp_has(g, p, 32))
__device__ __forceinline__ uint64_t grp_text32(const GrpText &g, uint32_t p)
{
    const uint32_t r = (uint32_t)((int32_t)p - g.base);
    lds_u32 *q = g.lds + (r >> 4);
    const uint32_t sh = 2u * (r & 15u);
    uint64_t x = (((uint64_t)q[1] << 32) | q[0]) >> sh;
    if (sh)
        x |= (uint64_t)q[2] << (64u - sh);
    return x;
}

// ... from anywhere in the text (p < n): the window, or the packed buffer itself
__device__ __forceinline__ uint64_t grp_text32_any(const GrpText &g, uint32_t p)
{
    if (grp_has(g, p, 32))
        return grp_text32(g, p);
    uint64_t P;
    uint32_t I;
    const uint32_t left = g.tv.len - p;
    sv_fetch32(g.tv, p, left < 32 ? left : 32, true, P, I);
    return P;
}

// the 32 symbols before text position i (i >= 1), field b = text[i - 1 - b] (fields of positions before the text: unspecified)
__device__ __forceinline__ uint64_t grp_back32_text(const GrpText &g, uint32_t i)
{
    if (i >= 32)
        return sv_rev2_64(grp_text32_any(g, i - 32));
    return sv_rev2_64(grp_text32_any(g, 0) << (2u * (32u - i)));
}
__device__ __forceinline__ uint64_t grp_back32_ref(glb_u32 *rw, uint32_t h)
{
    if (h >= 32)
        return sv_rev2_64(grp_raw64(rw, h - 32));
    return sv_rev2_64(grp_raw64(rw, 0) << (2u * (32u - h)));
}

// common prefix of text[tp ..) and ref[rp ..), at most max_len (bounded by both ends by the caller): 16 lanes x 32 symbols a step
// (refresh::matching_length as compare_fwd uses it, lz_diff.h:264-266)
__device__ uint32_t grp_match_fwd(GrpText &g, uint32_t tp, glb_u32 *rw, uint32_t rp, uint32_t max_len)
{
    const uint32_t gl = lane_id() & 15u;
    for (uint32_t base = 0;; base += GRP_STEP) {
        if (base < max_len && !grp_has(g, tp + base, GRP_STEP))
            grp_fill(g, tp + base);
        const uint32_t off = base + gl * 32u;
        bool stop = true;
        uint32_t so = 0;
        if (off < max_len) {
            const uint32_t cnt = max_len - off < 32u ? max_len - off : 32u;
            uint64_t d = grp_text32(g, tp + off) ^ grp_raw64(rw, (uint64_t)rp + off);
            if (cnt < 32u)
                d &= (1ULL << (2u * cnt)) - 1ULL;
            so = d ? ctz64(d) >> 1 : cnt;
            stop = so < 32u; // (a chunk cut by max_len stops too)
        }
        const uint32_t m = grp_ballot(stop);
        if (m) {
            const uint32_t l = (uint32_t)__builtin_ctz(m);
            return base + l * 32u + grp_bcast(so, l);
        }
    }
}

struct GrpRef {
    glb_u32 *words;
    glb_u32 *table; // (entries of 4 or 8 bytes)
    uint32_t ref_size, ht_mask, key_len, mml, is_short;
};

__device__ uint32_t lz_encode_group(const GrpRef &rd, GrpText &g, uint8_t *__restrict__ out)
{
    const uint32_t gl = lane_id() & 15u;
    const bool writer = gl == 0;
    const uint32_t n = g.tv.len;
    const uint32_t key_len = rd.key_len, mml = rd.mml, ref_size = rd.ref_size, ht_mask = rd.ht_mask;
    glb_u32 *rw = rd.words;

    // identical sequence (lz_diff.cpp:678-680)
    if (n == ref_size && grp_match_fwd(g, 0, rw, 0, n) == n)
        return 0;

    uint32_t i = 0, pred_pos = 0, npl = 0, o = 0;
    bool force_exact = false; // the grouped probe could not settle position i: 16 slots a round for this position
    while (i + key_len < n) {
        if (!grp_has(g, i, 36))
            grp_fill(g, i);
        // ---- grouped probe: positions i .. i+3, four slots each.  A position whose chain ends (an empty slot) within its four
        // slots without a fingerprint hit is a literal; the first position with hits before the empty slot goes to the
        // verification with exactly the candidates the reference's probe loop would visit, in its order; a chain longer than
        // four slots is left to the exact rounds below.
        uint32_t cand = 0, epos = 0;
        bool have_cands = false;
        if (!force_exact) {
            const uint32_t fq = gl >> 2, s = gl & 3u;
            const uint64_t P = grp_text32(g, i + fq);
            const uint64_t hx = murmur64(sv_key_from_packed(P, key_len));
            bool is_empty = false, fp_ok = false;
            if (i + fq + key_len < n) {
                const uint32_t sl = ((uint32_t)hx + s) & ht_mask;
                if (rd.is_short) {
                    const uint32_t e = rd.table[sl];
                    is_empty = e == 0xFFFFFFFFu;
                    epos = e >> 16;
                    fp_ok = (e & 0xFFFFu) == (uint32_t)(hx >> 48);
                } else {
                    const uint64_t e = ((glb_u64 *)rd.table)[sl];
                    is_empty = e == ~0ULL;
                    epos = (uint32_t)(e >> 32);
                    fp_ok = (uint32_t)e == (uint32_t)(hx >> 32);
                }
            }
            const uint32_t em = grp_ballot(is_empty), fm = grp_ballot(fp_ok && !is_empty);
            uint32_t f = 0;
            bool to_exact = false;
            for (; f < 4; ++f) {
                if (!(i + f + key_len < n))
                    break; // the loop ends here
                const uint32_t em4 = (em >> (4u * f)) & 15u;
                if (!em4) {
                    to_exact = true;
                    break;
                }
                const uint32_t c4 = (fm >> (4u * f)) & ((1u << __builtin_ctz(em4)) - 1u);
                if (c4) {
                    cand = c4 << (4u * f);
                    have_cands = true;
                    break;
                }
            }
            if (f) {
                // (every byte of the delta is stored by the group's lane 0: stores of one lane keep their order, nothing to
                // drain when a match rolls literals back or patches them)
                const uint32_t lits = (uint32_t)grp_bcast((uint32_t)P, 0); // position i's symbols (lane 0 of the group: fq == 0)
                if (writer)
                    for (uint32_t t = 0; t < f; ++t)
                        out[o + t] = (uint8_t)('A' + ((lits >> (2u * t)) & 3u));
                o += f;
                i += f;
                pred_pos += f;
                npl += f;
            }
            if (!have_cands) {
                force_exact = to_exact;
                if (f)
                    continue;
            }
        }
        const uint32_t max_len = n - i;
        uint32_t len_bck = 0, len_fwd = 0, match_pos = 0, min_to_update = mml;
        uint64_t best_x = 0; // text ^ reference over the 32 symbols before the chosen candidate (field b: distance b + 1)
        uint64_t hxe = 0;
        if (!have_cands) {
            force_exact = false;
            hxe = murmur64(sv_key_from_packed(grp_text32(g, i), key_len));
        }
        // ---- find_best_match (lz_diff.cpp:286-347): candidates in slot order up to the first empty slot / MAX_NO_TRIES probes
        for (uint32_t t = 0;; t += GRP) {
            bool last = true;
            if (!have_cands) {
                const uint32_t sl = ((uint32_t)hxe + t + gl) & ht_mask;
                bool is_empty, fp_ok;
                if (rd.is_short) {
                    const uint32_t e = rd.table[sl];
                    is_empty = e == 0xFFFFFFFFu;
                    epos = e >> 16;
                    fp_ok = (e & 0xFFFFu) == (uint32_t)(hxe >> 48);
                } else {
                    const uint64_t e = ((glb_u64 *)rd.table)[sl];
                    is_empty = e == ~0ULL;
                    epos = (uint32_t)(e >> 32);
                    fp_ok = (uint32_t)e == (uint32_t)(hxe >> 32);
                }
                const uint32_t em16 = grp_ballot(is_empty);
                cand = grp_ballot(fp_ok && !is_empty);
                if (em16)
                    cand &= (1u << __builtin_ctz(em16)) - 1u;
                last = em16 != 0 || t + GRP >= MAX_NO_TRIES;
            }
            while (cand) {
                const uint32_t j = (uint32_t)__builtin_ctz(cand);
                cand &= cand - 1;
                const uint32_t h_pos = grp_bcast(epos, j) * HASHING_STEP;
                const uint32_t lim = npl < h_pos ? npl : h_pos;
                // (the reference pads its copy with key_len symbols no text holds, lz_diff.cpp:48-53: a compare ends at the
                // reference's end at the latest -- here by the bound)
                const uint32_t ref_left = ref_size - h_pos;
                const uint32_t f_len = grp_match_fwd(g, i, rw, h_pos, max_len < ref_left ? max_len : ref_left);
                if (f_len >= key_len) {
                    // backward (lz_diff.cpp:308-311): 32 symbols a step, every lane the same words
                    uint32_t b_len = 0;
                    uint64_t x0 = 0;
                    if (lim) {
                        x0 = grp_back32_text(g, i) ^ grp_back32_ref(rw, h_pos);
                        const uint32_t v = lim < 32u ? lim : 32u;
                        const uint32_t e = x0 ? ctz64(x0) >> 1 : 32u;
                        b_len = e < v ? e : v;
                        if (b_len == 32u && lim > 32u) {
                            for (uint32_t k = 32; k < lim; k += 32) {
                                const uint64_t x = grp_back32_text(g, i - k) ^ grp_back32_ref(rw, h_pos - k);
                                const uint32_t vv = lim - k < 32u ? lim - k : 32u;
                                const uint32_t ee = x ? ctz64(x) >> 1 : 32u;
                                const uint32_t c = ee < vv ? ee : vv;
                                b_len += c;
                                if (c < 32u)
                                    break;
                            }
                        }
                    }
                    if (b_len + f_len > min_to_update) {
                        len_bck = b_len;
                        len_fwd = f_len;
                        match_pos = h_pos;
                        min_to_update = b_len + f_len;
                        best_x = x0;
                    }
                }
            }
            if (last)
                break;
        }

        if (len_bck + len_fwd < mml) {
            // literal
            const uint32_t s0 = (uint32_t)grp_text32(g, i) & 3u;
            if (writer)
                out[o] = (uint8_t)('A' + s0);
            ++o;
            ++i;
            ++pred_pos;
            ++npl;
            continue;
        }

        const uint32_t len = len_bck + len_fwd;
        // roll the back extension back (lz_diff.cpp:756-766)
        o -= len_bck;
        match_pos -= len_bck;
        pred_pos -= len_bck;
        i -= len_bck;
        const uint32_t n_trail = npl - len_bck; // literal bytes now ending the delta
        if (match_pos == pred_pos && n_trail) {
            // Literals equal to the reference become '!' (lz_diff.cpp:769-779): the reference walks back over the delta while
            // the bytes are letters (always, here), t < e_size, t < match_pos; byte o - t is text[i - t], i.e. distance
            // b + 1 = t + len_bck of the backward compare above
            uint32_t tmax = n_trail;
            if (o - 1u < tmax)
                tmax = o - 1u; // (o >= n_trail >= 1)
            if (match_pos - 1u < tmax || !match_pos)
                tmax = match_pos ? match_pos - 1u : 0u;
            const uint32_t in32 = len_bck < 32u ? 32u - len_bck : 0u; // values of t the 32 compared symbols cover
            const uint32_t t32 = tmax < in32 ? tmax : in32;
            if (t32) {
                // fields of best_x that are zero, as one bit per field, for b = len_bck .. len_bck + t32 - 1
                uint64_t z = ~(best_x | (best_x >> 1)) & 0x5555555555555555ULL;
                z >>= 2u * len_bck;
                if (t32 < 32u)
                    z &= (1ULL << (2u * t32)) - 1ULL;
                while (z) {
                    const uint32_t t = (ctz64(z) >> 1) + 1u;
                    z &= z - 1;
                    if (writer)
                        out[o - t] = '!';
                }
            }
            if (tmax > in32 && writer) // (a literal run longer than the compared symbols: rare)
                for (uint32_t tt = in32 + 1u; tt <= tmax; ++tt)
                    if (sv_sym(g.tv, i - tt, true) == ((rw[(match_pos - tt) >> 4] >> (2u * ((match_pos - tt) & 15u))) & 3u))
                        out[o - tt] = '!';
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
        pred_pos = match_pos + len;
        i += len;
        npl = 0;
    }
    // tail literals (lz_diff.cpp:795-796)
    if (i < n) {
        const uint32_t cnt = n - i;
        if (writer)
            for (uint32_t t = 0; t < cnt; ++t)
                out[o + t] = (uint8_t)('A' + sv_sym(g.tv, i + t, true));
        o += cnt;
    }
    return o;
}

// Group q of the launch parses segment q of the longest-first list; segments the group parse does not take (an escaped block in
// the text, a reference with symbols outside ACGT) are appended to defer_list for lz_parse_kernel<MODE_ENCODE>.
__global__ void __launch_bounds__(256) lz_encode_grp_kernel(const RefDesc *__restrict__ refs, const SegDesc *__restrict__ segs, uint32_t n_segs,
                                                            uint8_t *__restrict__ out_bytes, uint32_t *__restrict__ res_value,
                                                            const uint32_t *__restrict__ n_segs_dev, uint32_t *__restrict__ defer_list,
                                                            uint32_t *__restrict__ defer_count)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_win[256 / GRP][GW_WORDS + 4];
    const uint32_t grp = threadIdx.x / GRP, gl = threadIdx.x % GRP;
    const uint32_t idx = blockIdx.x * (256 / GRP) + grp;
    if (idx >= (n_segs_dev ? *n_segs_dev : n_segs))
        return;
    const SegDesc &sd = segs[idx];
    const RefDesc &rdm = refs[sd.ref_slot];
    GrpText g;
    g.tv = sd.text;
    g.lds = (lds_u32 *)&s_win[grp][0];
    g.base = 0;
    g.filled = false;
    bool dirty = rdm.esc_index != nullptr || rdm.key_len > 32u;
    if (!dirty && g.tv.esc_index && g.tv.len) {
        const uint64_t b0 = g.tv.start / SV_BLOCK, b1 = (g.tv.start + g.tv.len - 1) / SV_BLOCK;
        bool any = false;
        for (uint64_t b = b0 + gl; b <= b1; b += GRP)
            any = any || g.tv.esc_index[b] >= 0;
        dirty = grp_ballot(any) != 0;
    }
    if (dirty) {
        if (gl == 0)
            defer_list[atomicAdd(defer_count, 1u)] = idx;
        return;
    }
    if (gl < 4)
        g.lds[GW_WORDS + gl] = 0;
    GrpRef rd;
    rd.words = (glb_u32 *)rdm.words;
    rd.table = (glb_u32 *)rdm.table;
    rd.ref_size = rdm.ref_size;
    rd.ht_mask = rdm.ht_mask;
    rd.key_len = rdm.key_len;
    rd.mml = rdm.min_match_len;
    rd.is_short = rdm.is_short;
    const uint32_t v = lz_encode_group(rd, g, out_bytes + sd.out_off);
    if (gl == 0)
        res_value[sd.idx] = v;
}

} // namespace agc
