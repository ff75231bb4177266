/*
 * agc_oracle.c -- CPU restatement of AGC's segment-compression hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under agc_amd/ (the product) may import,
 * link or call this file.  It exists so that tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg can check / time the HIP path against a
 * plain, scalar statement of the reference algorithm.
 *
 * Parity pin: every function below is compared with the reference's own code
 * compiled in place from /root/reference into oracle/_ref/ (oracle/Makefile;
 * tests/test_oracle_golden.py::test_oracle_vs_reference_live), and with the
 * golden vectors committed under tests/golden/ that were generated from that
 * build (same file).  The whole-archive tests and the fuzzer
 * (tests/test_host_devsim.py, tests/test_fuzz_archives.py) exercise it further
 * through the device stand-in against the reference CLI.
 *
 * All citations are file:line under /root/reference/.
 * Written from the algorithm description; no reference
 * source text is reproduced.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

#define AGCO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------ */
/* a1: FASTA body bytes -> symbol codes                                     */
/* src/core/agc_compressor.cpp:907-951, table src/common/agc_basic.h:40-50  */
/* ------------------------------------------------------------------------ */
static u8 g_cnv[128];
static int g_cnv_ready = 0;

static void cnv_init(void)
{
    /* letters: A0 C1 G2 T3 N4 R5 Y6 S7 W8 K9 M10 B11 D12 H13 V14 U15,      */
    /* every other letter 30, '@' and '`' 32; bytes < 64 are dropped before  */
    /* the table is consulted.                                               */
    static const char *named = "ACGTNRYSWKMBDHVU";
    for (int c = 0; c < 128; ++c)
        g_cnv[c] = 30;
    for (int c = 0; c < 64; ++c)
        g_cnv[c] = 32; /* never consulted (dropped) */
    g_cnv[64] = 32;
    g_cnv[96] = 32;
    for (int i = 0; named[i]; ++i) {
        g_cnv[(int)named[i]] = (u8)i;
        g_cnv[(int)named[i] + 32] = (u8)i;
    }
    /* non-letters in 91..95 and 123..127 map to 30 in the reference table */
    g_cnv_ready = 1;
}

AGCO_API size_t agco_preprocess(const u8 *raw, size_t n, u8 *out)
{
    if (!g_cnv_ready)
        cnv_init();
    size_t o = 0;
    for (size_t i = 0; i < n; ++i) {
        u8 c = raw[i];
        if (c >> 6)
            out[o++] = g_cnv[c & 127];
    }
    return o;
}

/* src/common/agc_basic.cpp:282-315 */
AGCO_API void agco_rev_comp(const u8 *src, size_t n, u8 *dst)
{
    for (size_t i = 0; i < n; ++i) {
        u8 c = src[n - 1 - i];
        dst[i] = c < 4 ? (u8)(3 - c) : c;
    }
}

/* src/common/utils.h:164-176 */
static inline u64 murmur64(u64 h)
{
    h ^= h >> 33;
    h *= 0xff51afd7ed558ccdULL;
    h ^= h >> 33;
    h *= 0xc4ceb9fe1a85ec53ULL;
    h ^= h >> 33;
    return h;
}

AGCO_API u64 agco_murmur64(u64 h) { return murmur64(h); }

/* ------------------------------------------------------------------------ */
/* a2: rolling canonical k-mer.  src/core/kmer.h:284-301, 223-227, 360-362  */
/* dir and rc are both left-aligned in 64 bits.                              */
/* ------------------------------------------------------------------------ */
typedef struct {
    u64 dir, rc;
    u32 cur, k;
} kmer_t;

static inline void kmer_reset(kmer_t *km)
{
    km->dir = km->rc = 0;
    km->cur = 0;
}

static inline void kmer_insert(kmer_t *km, u64 sym)
{
    const u32 shift = 64 - 2 * km->k;
    const u64 mask = (~0ULL) << shift;
    km->rc >>= 2;
    km->rc += (3 - sym) << 62;
    km->rc &= mask;
    if (km->cur == km->k) {
        km->dir <<= 2;
        km->dir += sym << shift;
    } else {
        ++km->cur;
        km->dir += sym << (64 - 2 * km->cur);
    }
}

/* exact membership in a sorted u64 array (semantics of hs.h:489-497; the   */
/* bloom filter utils_adv.h:180-282 is a pure accelerator)                   */
static int in_sorted(const u64 *a, size_t n, u64 x)
{
    size_t lo = 0, hi = n;
    while (lo < hi) {
        size_t mid = (lo + hi) >> 1;
        if (a[mid] < x)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo < n && a[lo] == x;
}

/* ------------------------------------------------------------------------ */
/* a3: splitter scan of one contig.  src/core/agc_compressor.cpp:1997-2051  */
/* Emits the ordered list of segments; the k-mer is reset after every hit   */
/* so consecutive segments overlap by exactly k symbols.        */
/* Outputs (arrays sized by the caller, cap entries):                        */
/*   seg_start/seg_len, front/back k-mer (dir, rc) and full flags.           */
/* Returns the number of segments (may exceed cap: caller re-calls).         */
/* ------------------------------------------------------------------------ */
AGCO_API size_t agco_scan_contig(const u8 *ctg, size_t n, u32 k,
                                 const u64 *splitters_sorted, size_t n_spl,
                                 size_t cap, u64 *seg_start, u64 *seg_len,
                                 u64 *front_dir, u64 *front_rc, u8 *front_full,
                                 u64 *back_dir, u64 *back_rc, u8 *back_full)
{
    kmer_t km = {0, 0, 0, k};
    kmer_t split = {0, 0, 0, k};
    u64 split_pos = 0;
    size_t n_seg = 0;

    for (u64 pos = 0; pos < n; ++pos) {
        u8 x = ctg[pos];
        if (x >> 2) {
            kmer_reset(&km);
            continue;
        }
        kmer_insert(&km, x);
        if (km.cur != k)
            continue;
        u64 d = km.dir < km.rc ? km.dir : km.rc;
        if (!in_sorted(splitters_sorted, n_spl, d))
            continue;
        if (n_seg < cap) {
            seg_start[n_seg] = split_pos;
            seg_len[n_seg] = pos + 1 - split_pos;
            front_dir[n_seg] = split.dir;
            front_rc[n_seg] = split.rc;
            front_full[n_seg] = split.cur == k;
            back_dir[n_seg] = km.dir;
            back_rc[n_seg] = km.rc;
            back_full[n_seg] = 1;
        }
        ++n_seg;
        split_pos = pos + 1 - k;
        split = km;
        kmer_reset(&km);
    }
    if (split_pos < n) {
        if (n_seg < cap) {
            seg_start[n_seg] = split_pos;
            seg_len[n_seg] = n - split_pos;
            front_dir[n_seg] = split.dir;
            front_rc[n_seg] = split.rc;
            front_full[n_seg] = split.cur == k;
            back_dir[n_seg] = 0;
            back_rc[n_seg] = 0;
            back_full[n_seg] = 0;
        }
        ++n_seg;
    }
    return n_seg;
}

/* ------------------------------------------------------------------------ */
/* Reference preprocessing (needed to make splitter sets for tests).         */
/* enumerate canonical k-mers: agc_compressor.cpp:630-660                    */
/* ------------------------------------------------------------------------ */
AGCO_API size_t agco_enumerate_kmers(const u8 *ctg, size_t n, u32 k, u64 *out)
{
    kmer_t km = {0, 0, 0, k};
    size_t m = 0;
    for (size_t i = 0; i < n; ++i) {
        u8 x = ctg[i];
        if (x > 3) {
            kmer_reset(&km);
            continue;
        }
        kmer_insert(&km, x);
        if (km.cur == k)
            out[m++] = km.dir < km.rc ? km.dir : km.rc;
    }
    return m;
}

static int cmp_u64(const void *a, const void *b)
{
    u64 x = *(const u64 *)a, y = *(const u64 *)b;
    return x < y ? -1 : x > y;
}

/* sort + keep singletons: agc_compressor.cpp:482-491, 664-680 */
AGCO_API size_t agco_sort_keep_singletons(u64 *v, size_t n)
{
    qsort(v, n, sizeof(u64), cmp_u64);
    size_t o = 0;
    for (size_t i = 0; i < n;) {
        size_t j = i + 1;
        while (j < n && v[j] == v[i])
            ++j;
        if (j == i + 1)
            v[o++] = v[i];
        i = j;
    }
    return o;
}

/* splitters of one reference contig: agc_compressor.cpp:762-825 (fallback   */
/* minimizers are dead code at the default -f 0).             */
AGCO_API size_t agco_find_splitters_in_contig(const u8 *ctg, size_t n, u32 k,
                                              u64 segment_size,
                                              const u64 *singletons, size_t n_sing,
                                              u64 *out)
{
    kmer_t km = {0, 0, 0, k};
    u64 current_len = segment_size;
    size_t n_out = 0;
    /* recent k-mers since the last splitter: remember as [first,last] index */
    /* range of positions; re-enumerate at the end instead of storing them.  */
    size_t recent_from = 0; /* symbol index right after the last reset */

    for (size_t i = 0; i < n; ++i) {
        u8 x = ctg[i];
        if (x > 3)
            kmer_reset(&km);
        else {
            kmer_insert(&km, x);
            if (km.cur == k && current_len >= segment_size) {
                u64 d = km.dir < km.rc ? km.dir : km.rc;
                if (in_sorted(singletons, n_sing, d)) {
                    out[n_out++] = d;
                    current_len = 0;
                    kmer_reset(&km);
                    recent_from = i + 1;
                }
            }
        }
        ++current_len;
    }
    /* right-most singleton among the k-mers seen since the last splitter */
    {
        kmer_t t = {0, 0, 0, k};
        u64 best = 0;
        int have = 0;
        for (size_t i = recent_from; i < n; ++i) {
            u8 x = ctg[i];
            if (x > 3) {
                kmer_reset(&t);
                continue;
            }
            kmer_insert(&t, x);
            if (t.cur == k) {
                u64 d = t.dir < t.rc ? t.dir : t.rc;
                if (in_sorted(singletons, n_sing, d)) {
                    best = d;
                    have = 1;
                }
            }
        }
        if (have)
            out[n_out++] = best;
    }
    return n_out;
}

/* ------------------------------------------------------------------------ */
/* a10-a11, a6, a7: LZ-diff.  src/common/lz_diff.{h,cpp}                     */
/* ------------------------------------------------------------------------ */
typedef struct {
    u8 *ref;       /* reference + key_len bytes of 31 (lz_diff.cpp:48-53) */
    u32 ref_size;  /* unpadded */
    u32 min_match_len, key_len;
    u64 key_mask;
    u64 ht_size, ht_mask;
    int short_ht;  /* lz_diff.cpp:146 */
    u16 *ht16;
    u32 *ht32;
    int index_ready;
} lz_t;

enum { HASHING_STEP = 4, MAX_NO_TRIES = 64, INVALID_SYMBOL = 31, N_CODE = 4, N_RUN_STARTER = 30, MIN_NRUN = 4 };

/* lz_diff.h:58-106 */
static u64 get_code(const lz_t *z, const u8 *s)
{
    u64 x = 0;
    for (u32 i = 0; i < z->key_len; ++i) {
        if (s[i] > 3)
            return ~0ULL;
        x = (x << 2) + s[i];
    }
    return x;
}

/* lz_diff.cpp:81-141 sizing, :375-428 insertion */
static void lz_prepare_index(lz_t *z)
{
    const u32 padded = z->ref_size + z->key_len;
    u64 cnt = 0;
    u32 no_prev_valid = 0, cnt_mod = 0;
    const u32 key_len_mod = z->key_len % HASHING_STEP;
    for (u32 j = 0; j < padded; ++j) {
        if (z->ref[j] < 4)
            ++no_prev_valid;
        else
            no_prev_valid = 0;
        if (++cnt_mod == HASHING_STEP)
            cnt_mod = 0;
        if (cnt_mod == key_len_mod && no_prev_valid >= z->key_len)
            ++cnt;
    }
    u64 hs = (u64)((double)cnt / 0.7);
    while (hs & (hs - 1))
        hs &= hs - 1;
    hs <<= 1;
    if (hs < 8)
        hs = 8;
    z->ht_size = hs;
    z->ht_mask = hs - 1;
    if (z->short_ht) {
        z->ht16 = (u16 *)malloc(hs * sizeof(u16));
        memset(z->ht16, 0xff, hs * sizeof(u16));
    } else {
        z->ht32 = (u32 *)malloc(hs * sizeof(u32));
        memset(z->ht32, 0xff, hs * sizeof(u32));
    }
    for (u32 i = 0; i + z->key_len < padded; i += HASHING_STEP) {
        u64 x = get_code(z, z->ref + i);
        if (x == ~0ULL)
            continue;
        u64 pos = murmur64(x) & z->ht_mask;
        for (u32 j = 0; j < MAX_NO_TRIES; ++j) {
            u64 p = (pos + j) & z->ht_mask;
            if (z->short_ht) {
                if (z->ht16[p] == 0xffff) {
                    z->ht16[p] = (u16)(i / HASHING_STEP);
                    break;
                }
            } else {
                if (z->ht32[p] == 0xffffffffu) {
                    z->ht32[p] = i / HASHING_STEP;
                    break;
                }
            }
        }
    }
    z->index_ready = 1;
}

AGCO_API void *agco_lz_create(const u8 *ref, u32 n, u32 min_match_len)
{
    lz_t *z = (lz_t *)calloc(1, sizeof(lz_t));
    z->min_match_len = min_match_len;
    z->key_len = min_match_len - HASHING_STEP + 1; /* lz_diff.cpp:19 */
    z->key_mask = ~0ULL >> (64 - 2 * z->key_len);
    z->ref_size = n;
    z->short_ht = (n / HASHING_STEP) < 65535;
    z->ref = (u8 *)malloc((size_t)n + z->key_len + 64);
    memcpy(z->ref, ref, n);
    memset(z->ref + n, INVALID_SYMBOL, z->key_len + 64);
    return z;
}

AGCO_API void agco_lz_free(void *h)
{
    lz_t *z = (lz_t *)h;
    if (!z)
        return;
    free(z->ref);
    free(z->ht16);
    free(z->ht32);
    free(z);
}

/* index dump for parity checks of the device index build */
AGCO_API u64 agco_lz_index(void *h, int *is16, const void **table)
{
    lz_t *z = (lz_t *)h;
    if (!z->index_ready)
        lz_prepare_index(z);
    *is16 = z->short_ht;
    *table = z->short_ht ? (const void *)z->ht16 : (const void *)z->ht32;
    return z->ht_size;
}

static inline u32 common_prefix(const u8 *p, const u8 *q, u32 max_len)
{
    u32 l = 0;
    while (l < max_len && p[l] == q[l])
        ++l;
    return l;
}

/* lz_diff.cpp:287-372 (16- and 32-bit tables share one body here) */
static int find_best_match(const lz_t *z, u32 ht_pos, const u8 *s, u32 max_len,
                           u32 no_prev_literals, u32 *ref_pos, u32 *len_bck, u32 *len_fwd)
{
    *len_fwd = 0;
    *len_bck = 0;
    u32 min_to_update = z->min_match_len;
    for (u32 t = 0; t < MAX_NO_TRIES; ++t) {
        u32 e;
        if (z->short_ht) {
            if (z->ht16[ht_pos] == 0xffff)
                break;
            e = z->ht16[ht_pos];
        } else {
            if (z->ht32[ht_pos] == 0xffffffffu)
                break;
            e = z->ht32[ht_pos];
        }
        u32 h_pos = e * HASHING_STEP;
        const u8 *p = z->ref + h_pos;
        u32 f_len = common_prefix(s, p, max_len);
        if (f_len >= z->key_len) {
            u32 lim = no_prev_literals < h_pos ? no_prev_literals : h_pos;
            u32 b_len = 0;
            for (; b_len < lim; ++b_len)
                if (s[-(int64_t)b_len - 1] != p[-(int64_t)b_len - 1])
                    break;
            if (b_len + f_len > min_to_update) {
                *len_bck = b_len;
                *len_fwd = f_len;
                *ref_pos = h_pos;
                min_to_update = b_len + f_len;
            }
        }
        ht_pos = (u32)((ht_pos + 1u) & z->ht_mask);
    }
    return *len_bck + *len_fwd >= z->min_match_len;
}

/* lz_diff.h:122-132 */
static u32 nrun_len(const u8 *s, u32 max_len)
{
    if (s[0] != N_CODE || s[1] != N_CODE || s[2] != N_CODE)
        return 0;
    u32 len = 3;
    while (len < max_len && s[len] == N_CODE)
        ++len;
    return len;
}

/* lz_diff.h:229-262: decimal, optional '-' */
static size_t put_int(u8 *out, size_t o, int64_t x)
{
    char tmp[24];
    int n = 0;
    if (x == 0) {
        out[o++] = '0';
        return o;
    }
    if (x < 0) {
        out[o++] = '-';
        x = -x;
    }
    while (x) {
        tmp[n++] = (char)('0' + x % 10);
        x /= 10;
    }
    while (n)
        out[o++] = (u8)tmp[--n];
    return o;
}

static int text_equals_ref(const lz_t *z, const u8 *text, u32 n)
{
    return n == z->ref_size && memcmp(text, z->ref, n) == 0;
}

/*
 * CLZDiff_V2::Encode, lz_diff.cpp:669-798.  `out` must hold at least
 * n + 5*n/16 + 64 bytes (a match of >= 16 symbols costs at most 21 bytes).
 * Returns the encoded length.
 * The text buffer must be readable for key_len bytes before `text` is not
 * required; reads stay inside [text, text+n).
 */
AGCO_API size_t agco_lz_encode(void *h, const u8 *text, u32 n, u8 *out)
{
    lz_t *z = (lz_t *)h;
    if (!z->index_ready)
        lz_prepare_index(z);
    if (text_equals_ref(z, text, n))
        return 0;
    size_t o = 0;
    u32 i = 0, pred_pos = 0, no_prev_literals = 0;
    const u32 key_len = z->key_len;

    while (i + key_len < n) {
        const u8 *tp = text + i;
        u64 x = get_code(z, tp);
        if (x == ~0ULL) {
            u32 nr = nrun_len(tp, n - i);
            if (nr >= MIN_NRUN) {
                out[o++] = N_RUN_STARTER;
                o = put_int(out, o, (int64_t)nr - MIN_NRUN);
                out[o++] = N_CODE;
                i += nr;
                no_prev_literals = 0;
            } else {
                out[o++] = (u8)('A' + *tp);
                ++i;
                ++pred_pos;
                ++no_prev_literals;
            }
            continue;
        }
        u32 ht_pos = (u32)(murmur64(x) & z->ht_mask);
        u32 len_bck = 0, len_fwd = 0, match_pos = 0;
        u32 max_len = n - i;
        if (!find_best_match(z, ht_pos, tp, max_len, no_prev_literals, &match_pos, &len_bck, &len_fwd)) {
            out[o++] = (u8)('A' + *tp);
            ++i;
            ++pred_pos;
            ++no_prev_literals;
            continue;
        }
        if (len_bck) {
            o -= len_bck;
            match_pos -= len_bck;
            pred_pos -= len_bck;
            i -= len_bck;
        }
        if (match_pos == pred_pos) {
            /* lz_diff.cpp:769-779: literals equal to the reference -> '!' */
            u32 e_size = (u32)o;
            for (u32 t = 1; t < e_size && t < match_pos; ++t) {
                u8 c = out[e_size - t];
                if (c < 'A' || c > 'Z')
                    break;
                if ((u8)(c - 'A') == z->ref[match_pos - t])
                    out[e_size - t] = '!';
            }
        }
        u32 len = len_bck + len_fwd;
        int to_end = (i + len == n) && (match_pos + len == z->ref_size);
        o = put_int(out, o, (int64_t)(int)match_pos - (int64_t)(int)pred_pos);
        if (!to_end) {
            out[o++] = ',';
            o = put_int(out, o, (int64_t)len - z->min_match_len);
        }
        out[o++] = '.';
        pred_pos = match_pos + len;
        i += len;
        no_prev_literals = 0;
    }
    for (; i < n; ++i)
        out[o++] = (u8)('A' + text[i]);
    return o;
}

/* CLZDiff_V2 cost helpers, lz_diff.h:375-424 */
static u32 v2_uint_len(u32 x)
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
static u32 v2_int_len(int x)
{
    return x >= 0 ? v2_uint_len((u32)x) : 1 + v2_uint_len((u32)-x);
}
static u32 v2_cost_match(const lz_t *z, u32 ref_pos, u32 len, u32 pred_pos)
{
    int dif = (int)ref_pos - (int)pred_pos;
    u32 r = v2_int_len(dif);
    if (len != ~0u)
        r += 1 + v2_uint_len(len - z->min_match_len);
    return r + 1;
}

/*
 * CLZDiff_V2::Estimate, lz_diff.cpp:839-946.  Mirrors Encode but never rolls
 * the back-extension back (i, pred_pos and match_pos keep their un-shifted
 * values) and returns as soon as est_cost > bound at a loop top.  All
 * arithmetic is u32 and may wrap exactly as in the reference.
 * `peak` (optional) receives the largest est_cost seen at a loop-top check,
 * so that a caller can replay any bound from one unbounded run.
 */
AGCO_API u32 agco_lz_estimate(void *h, const u8 *text, u32 n, u32 bound, u32 *peak)
{
    lz_t *z = (lz_t *)h;
    if (!z->index_ready)
        lz_prepare_index(z);
    u32 pk = 0;
    if (peak)
        *peak = 0;
    if (text_equals_ref(z, text, n))
        return 0;
    u32 est = 0;
    u32 i = 0, pred_pos = 0, no_prev_literals = 0;
    const u32 key_len = z->key_len;

    while (i + key_len < n) {
        if (est > pk)
            pk = est;
        if (est > bound) {
            if (peak)
                *peak = pk;
            return est;
        }
        const u8 *tp = text + i;
        u64 x = get_code(z, tp);
        if (x == ~0ULL) {
            u32 nr = nrun_len(tp, n - i);
            if (nr >= MIN_NRUN) {
                est += 2 + v2_uint_len(nr); /* lz_diff.h:407-410: len, not len-4 */
                i += nr;
                no_prev_literals = 0;
            } else {
                ++est;
                ++i;
                ++pred_pos;
                ++no_prev_literals;
            }
            continue;
        }
        u32 ht_pos = (u32)(murmur64(x) & z->ht_mask);
        u32 len_bck = 0, len_fwd = 0, match_pos = 0;
        u32 max_len = n - i;
        if (!find_best_match(z, ht_pos, tp, max_len, no_prev_literals, &match_pos, &len_bck, &len_fwd)) {
            ++est;
            ++i;
            ++pred_pos;
            ++no_prev_literals;
            continue;
        }
        u32 len = len_bck + len_fwd;
        if (i + len == n && match_pos + len == z->ref_size)
            est += v2_cost_match(z, match_pos, ~0u, pred_pos);
        else
            est += v2_cost_match(z, match_pos, len, pred_pos);
        pred_pos = match_pos + len;
        i += len;
        no_prev_literals = 0;
    }
    est += n - i;
    if (peak)
        *peak = pk;
    return est;
}

/* base-class cost helpers, lz_diff.h:159-191 */
static u32 base_int_len(u32 x)
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
static u32 base_cost_match(const lz_t *z, u32 ref_pos, u32 len, u32 pred_pos)
{
    int dif = (int)ref_pos - (int)pred_pos;
    u32 r = dif >= 0 ? base_int_len((u32)dif) : base_int_len((u32)-dif) + 1;
    return r + base_int_len(len - z->min_match_len) + 2;
}

/*
 * CLZDiffBase::GetCodingCostVector, lz_diff.cpp:159-284.  Writes exactly n
 * costs.  Back-extension pops the literal costs already emitted.
 */
AGCO_API void agco_lz_cost_vector(void *h, const u8 *text, u32 n, int prefix_costs, u32 *costs)
{
    lz_t *z = (lz_t *)h;
    if (!z->index_ready)
        lz_prepare_index(z);
    size_t o = 0;
    u32 i = 0, pred_pos = 0, no_prev_literals = 0;
    const u32 key_len = z->key_len;

    while (i + key_len < n) {
        const u8 *tp = text + i;
        u64 x = get_code(z, tp);
        if (x == ~0ULL) {
            u32 nr = nrun_len(tp, n - i);
            if (nr >= MIN_NRUN) {
                u32 tc = 2 + base_int_len(nr - MIN_NRUN);
                if (prefix_costs)
                    costs[o++] = tc;
                for (u32 t = 0; t + 1 < nr; ++t)
                    costs[o++] = 0;
                if (!prefix_costs)
                    costs[o++] = tc;
                i += nr;
                no_prev_literals = 0;
            } else {
                costs[o++] = 1;
                ++i;
                ++pred_pos;
                ++no_prev_literals;
            }
            continue;
        }
        u32 ht_pos = (u32)(murmur64(x) & z->ht_mask);
        u32 len_bck = 0, len_fwd = 0, match_pos = 0;
        u32 max_len = n - i;
        if (!find_best_match(z, ht_pos, tp, max_len, no_prev_literals, &match_pos, &len_bck, &len_fwd)) {
            costs[o++] = 1;
            ++i;
            ++pred_pos;
            ++no_prev_literals;
            continue;
        }
        if (len_bck) {
            o -= len_bck;
            match_pos -= len_bck;
            pred_pos -= len_bck;
            i -= len_bck;
        }
        u32 len = len_bck + len_fwd;
        u32 tc = base_cost_match(z, match_pos, len, pred_pos);
        if (prefix_costs)
            costs[o++] = tc;
        for (u32 t = 0; t + 1 < len; ++t)
            costs[o++] = 0;
        if (!prefix_costs)
            costs[o++] = tc;
        pred_pos = match_pos + len;
        i += len;
        no_prev_literals = 0;
    }
    for (; i < n; ++i)
        costs[o++] = 1;
}

/*
 * CLZDiff_V2::Decode, lz_diff.cpp:801-836 (used for round-trip properties).
 * Returns decoded length (<= cap written).
 */
AGCO_API size_t agco_lz_decode(const u8 *ref, u32 ref_size, u32 min_match_len,
                               const u8 *enc, size_t enc_len, u8 *out, size_t cap)
{
    size_t o = 0, p = 0;
    u32 pred_pos = 0;
    while (p < enc_len) {
        u8 c = enc[p];
        if ((c >= 'A' && c <= 'A' + 20) || c == '!') {
            u8 s = c == '!' ? ref[pred_pos] : (u8)(c - 'A');
            if (o < cap)
                out[o] = s;
            ++o;
            ++pred_pos;
            ++p;
        } else if (c == N_RUN_STARTER) {
            ++p;
            int64_t v = 0;
            while (p < enc_len && enc[p] >= '0' && enc[p] <= '9')
                v = v * 10 + (enc[p++] - '0');
            ++p; /* N_CODE terminator */
            u32 len = (u32)(v + MIN_NRUN);
            for (u32 t = 0; t < len; ++t, ++o)
                if (o < cap)
                    out[o] = N_CODE;
        } else {
            int neg = 0;
            int64_t v = 0;
            if (enc[p] == '-') {
                neg = 1;
                ++p;
            }
            while (p < enc_len && enc[p] >= '0' && enc[p] <= '9')
                v = v * 10 + (enc[p++] - '0');
            if (neg)
                v = -v;
            u32 ref_pos = (u32)(v + (int64_t)pred_pos);
            u32 len;
            if (enc[p] == ',') {
                ++p;
                int64_t l = 0;
                while (p < enc_len && enc[p] >= '0' && enc[p] <= '9')
                    l = l * 10 + (enc[p++] - '0');
                len = (u32)(l + min_match_len);
            } else
                len = ref_size - ref_pos;
            ++p; /* '.' */
            for (u32 t = 0; t < len; ++t, ++o)
                if (o < cap)
                    out[o] = ref[ref_pos + t];
            pred_pos = ref_pos + len;
        }
    }
    return o;
}

/* ------------------------------------------------------------------------ */
/* a13: reference storage helpers.  src/common/segment.h:73-138, 218-255     */
/* ------------------------------------------------------------------------ */

/* repetitiveness probe: returns 1 if best_frac >= 0.5 (store raw, zstd 19), */
/* 0 if tuples + zstd 13.  Exact double arithmetic of segment.h:224-254.     */
AGCO_API int agco_ref_is_repetitive(const u8 *data, size_t n)
{
    double best_frac = 0.0;
    for (u32 lag = 4; lag < 32; ++lag) {
        u32 cnt = 0, cur = 0;
        for (u32 j = 0; (size_t)j + lag < n; ++j) {
            cnt += data[j] == data[(size_t)j + lag];
            cur += data[j] < 4;
        }
        double frac = 0.0;
        if (cur)
            frac = (double)cnt / cur;
        if (frac > best_frac) {
            best_frac = frac;
            if (best_frac >= 0.5)
                break;
        }
    }
    return !(best_frac < 0.5);
}

/* per-lag counters of the probe (what the device kernel returns) */
AGCO_API void agco_ref_lag_counts(const u8 *data, size_t n, u32 *cnt28, u32 *cur28)
{
    for (u32 lag = 4; lag < 32; ++lag) {
        u32 cnt = 0, cur = 0;
        for (u32 j = 0; (size_t)j + lag < n; ++j) {
            cnt += data[j] == data[(size_t)j + lag];
            cur += data[j] < 4;
        }
        cnt28[lag - 4] = cnt;
        cur28[lag - 4] = cur;
    }
}

/* bytes2tuples, segment.h:73-138.  out cap >= n + 2.  Returns tuple length. */
AGCO_API size_t agco_bytes2tuples(const u8 *v, size_t n, u8 *out)
{
    u8 me = 0;
    for (size_t i = 0; i < n; ++i)
        if (v[i] > me)
            me = v[i];
    u32 nb, mult;
    if (me < 4) {
        nb = 4;
        mult = 4;
    } else if (me < 6) {
        nb = 3;
        mult = 6;
    } else if (me < 16) {
        nb = 2;
        mult = 16;
    } else {
        memcpy(out, v, n);
        out[n] = 0x10;
        return n + 1;
    }
    size_t i = 0, o = 0;
    for (; i + nb <= n; i += nb) {
        u8 c = 0;
        for (u32 j = 0; j < nb; ++j)
            c = (u8)(c * mult + v[i + j]);
        out[o++] = c;
    }
    u8 c = 0;
    for (; i < n; ++i)
        c = (u8)(c * mult + v[i]);
    out[o++] = c;
    out[o++] = (u8)((nb << 4) + (n % nb));
    return o;
}
