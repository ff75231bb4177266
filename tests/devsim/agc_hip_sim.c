/*
 * agc_hip_sim.c -- a CPU stand-in for the device behind include/agc_hip.h.
 *
 * TEST INFRASTRUCTURE ONLY (tests/devsim/).  It implements the C ABI of the HIP library on top of the
 * CPU oracle (oracle/agc_oracle.c) so that the HOST side of the create path -- agc_amd/csrc/host/:
 * registration order, group bookkeeping, pack/zstd streams, collection metadata, container -- can be
 * exercised by the `-m "not gpu"` tests on a machine without a GPU: the host sources are compiled
 * unchanged and linked against this file instead of libagc_hip.so, into tests/devsim/_build/.
 * Nothing under agc_amd/ builds, links or loads it; agc_amd.build never produces it; the product
 * libraries keep failing with AGC_HIP_ENODEV without a HIP device.  "Device" pointers here are plain
 * host pointers.  It is also not a performance statement of any kind.
 *
 * The ABI contracts restated here are the ones written in include/agc_hip.h; the algorithms are the
 * oracle's (each cites the reference file:line there).  The split-point arithmetic follows
 * src/core/agc_compressor.cpp:1538-1617 directly.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/agc_hip.h"

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;

/* oracle entry points (oracle/agc_oracle.c) */
void agco_rev_comp(const u8 *src, size_t n, u8 *dst);
size_t agco_scan_contig(const u8 *ctg, size_t n, u32 k, const u64 *spl, size_t n_spl, size_t cap, u64 *seg_start, u64 *seg_len,
                        u64 *front_dir, u64 *front_rc, u8 *front_full, u64 *back_dir, u64 *back_rc, u8 *back_full);
size_t agco_enumerate_kmers(const u8 *ctg, size_t n, u32 k, u64 *out);
size_t agco_find_splitters_in_contig(const u8 *ctg, size_t n, u32 k, u64 segment_size, const u64 *sing, size_t n_sing, u64 *out);
void *agco_lz_create(const u8 *ref, u32 n, u32 min_match_len);
void agco_lz_free(void *h);
size_t agco_lz_encode(void *h, const u8 *text, u32 n, u8 *out);
u32 agco_lz_estimate(void *h, const u8 *text, u32 n, u32 bound, u32 *peak);
void agco_lz_cost_vector(void *h, const u8 *text, u32 n, int prefix_costs, u32 *costs);
void agco_ref_lag_counts(const u8 *data, size_t n, u32 *cnt28, u32 *cur28);

struct agc_hip_ctx {
    char err[256];
    u8 *sample;
    u64 sample_cap;
    u64 *spl; /* sorted */
    u64 n_spl;
    void **lz; /* by gid */
    u32 n_lz;
    /* encode in two halves: parsed at begin (see agc_hip_lz_encode_begin_dev), handed out by end */
    struct {
        int pending;
        u32 n;
        u8 *out;
        u64 *eoff;
    } enc[AGC_HIP_ENCODE_LANES]; /* one per lane (agc_hip_lz_encode_*_on) */
    /* the next sample ahead of its turn: the identity of what was announced */
    const void *pf_words;
    /* what agc_hip_segments_packed left for agc_hip_segments_encode_known */
    int sg_valid;
    u32 sg_ne, *sg_gid, *sg_len;
    u64 *sg_off;
    u8 *sg_rc, *sg_codes;
    /* the mirrored (k-mer 1, k-mer 2) -> group table */
    agc_hip_group_slot *gmap;
    u64 gmap_slots;
    /* agc_hip_pack_fasta_begin .. _end: the result of the one conversion in flight */
    int pfa_pending, pfa_rc;
    u32 pfa_n_ctg;
    u64 *pfa_off, pfa_esc;
    /* agc_hip_sample_pack: the context's own packed buffers */
    uint32_t *sp_words;
    int32_t *sp_index;
    u8 *sp_esc;
};

static int cmp_u64(const void *a, const void *b)
{
    u64 x = *(const u64 *)a, y = *(const u64 *)b;
    return x < y ? -1 : x > y;
}

static size_t sort_unique(u64 *v, size_t n)
{
    qsort(v, n, sizeof(u64), cmp_u64);
    size_t o = 0;
    for (size_t i = 0; i < n; ++i)
        if (!o || v[o - 1] != v[i])
            v[o++] = v[i];
    return o;
}

static int fail(agc_hip_ctx *c, int code, const char *msg)
{
    if (c)
        snprintf(c->err, sizeof c->err, "devsim: %s", msg);
    return code;
}

/* slice as the kernels read it: reverse-complemented when rc */
static u8 *slice(const u8 *base, u64 off, u32 len, int rc)
{
    u8 *t = (u8 *)malloc((size_t)len + 64);
    if (rc)
        agco_rev_comp(base + off, len, t);
    else
        memcpy(t, base + off, len);
    return t;
}

int agc_hip_create(agc_hip_ctx **out, int device)
{
    (void)device;
    if (!out)
        return AGC_HIP_EINVAL;
    *out = (agc_hip_ctx *)calloc(1, sizeof(agc_hip_ctx));
    return *out ? AGC_HIP_OK : AGC_HIP_ENOMEM;
}

void agc_hip_destroy(agc_hip_ctx *c)
{
    if (!c)
        return;
    for (u32 i = 0; i < c->n_lz; ++i)
        agco_lz_free(c->lz[i]);
    free(c->lz);
    free(c->spl);
    free(c->sample);
    free(c->sp_words);
    free(c->sp_index);
    free(c->sp_esc);
    free(c->gmap);
    free(c->pfa_off);
    free(c->sg_gid), free(c->sg_len), free(c->sg_off), free(c->sg_rc), free(c->sg_codes);
    for (u32 l = 0; l < AGC_HIP_ENCODE_LANES; ++l)
        free(c->enc[l].out), free(c->enc[l].eoff);
    free(c);
}

const char *agc_hip_last_error(const agc_hip_ctx *c) { return c ? c->err : "no context"; }
uint32_t agc_hip_abi_version(void) { return 0x5157u; /* not the product's */ }
int agc_hip_sync(agc_hip_ctx *c) { return c ? AGC_HIP_OK : AGC_HIP_EINVAL; }
int agc_hip_timing_enable(agc_hip_ctx *c, int on) { (void)c; (void)on; return AGC_HIP_OK; }
int agc_hip_timing_reset(agc_hip_ctx *c) { (void)c; return AGC_HIP_OK; }
int agc_hip_timing_get(agc_hip_ctx *c, int which, double *ms, uint64_t *launches)
{
    (void)c; (void)which;
    if (ms) *ms = 0;
    if (launches) *launches = 0;
    return AGC_HIP_OK;
}

int agc_hip_sample_buffer(agc_hip_ctx *c, uint64_t bytes, uint8_t **d_ptr)
{
    if (!c || !d_ptr)
        return AGC_HIP_EINVAL;
    if (bytes + 4096 > c->sample_cap) {
        free(c->sample);
        c->sample_cap = bytes + bytes / 4 + 4096;
        c->sample = (u8 *)malloc(c->sample_cap);
        if (!c->sample)
            return fail(c, AGC_HIP_ENOMEM, "sample buffer");
    }
    *d_ptr = c->sample;
    return AGC_HIP_OK;
}

int agc_hip_copy_to_device(agc_hip_ctx *c, uint8_t *d_dst, const uint8_t *h_src, uint64_t n)
{
    if (!c || (n && (!d_dst || !h_src)))
        return AGC_HIP_EINVAL;
    memcpy(d_dst, h_src, n);
    return AGC_HIP_OK;
}

int agc_hip_splitters_set(agc_hip_ctx *c, const uint64_t *h, uint64_t n)
{
    if (!c || (n && !h))
        return AGC_HIP_EINVAL;
    free(c->spl);
    c->spl = (u64 *)malloc((n + 1) * sizeof(u64));
    memcpy(c->spl, h, n * sizeof(u64));
    c->n_spl = sort_unique(c->spl, n);
    return AGC_HIP_OK;
}

int agc_hip_splitters_insert(agc_hip_ctx *c, const uint64_t *h, uint64_t n)
{
    if (!c || (n && !h))
        return AGC_HIP_EINVAL;
    c->spl = (u64 *)realloc(c->spl, (c->n_spl + n + 1) * sizeof(u64));
    memcpy(c->spl + c->n_spl, h, n * sizeof(u64));
    c->n_spl = sort_unique(c->spl, c->n_spl + n);
    return AGC_HIP_OK;
}

uint64_t agc_hip_splitters_count(const agc_hip_ctx *c) { return c ? c->n_spl : 0; }

int agc_hip_determine_splitters_dev(agc_hip_ctx *c, const uint8_t *d, const uint64_t *off, uint32_t n_ctg, uint32_t k, uint32_t segment_size,
                                    uint64_t cap, uint64_t *h_spl, uint64_t *h_n_spl, uint64_t sorted_cap, uint64_t *h_sorted,
                                    uint64_t *h_n_sorted)
{
    if (!c || !h_n_spl || (n_ctg && (!d || !off)))
        return AGC_HIP_EINVAL;
    const u64 tot = n_ctg ? off[n_ctg] - off[0] : 0;
    u64 *all = (u64 *)malloc((tot + 1) * sizeof(u64));
    size_t n_all = 0;
    for (u32 i = 0; i < n_ctg; ++i)
        n_all += agco_enumerate_kmers(d + off[i], off[i + 1] - off[i], k, all + n_all);
    qsort(all, n_all, sizeof(u64), cmp_u64);
    if (h_sorted) {
        if (h_n_sorted)
            *h_n_sorted = n_all;
        if (n_all > sorted_cap) {
            free(all);
            return fail(c, AGC_HIP_ECAP, "sorted k-mer buffer");
        }
        memcpy(h_sorted, all, n_all * sizeof(u64));
    }
    u64 *sing = (u64 *)malloc((n_all + 1) * sizeof(u64));
    size_t n_sing = 0;
    for (size_t i = 0; i < n_all;) {
        size_t j = i + 1;
        while (j < n_all && all[j] == all[i])
            ++j;
        if (j == i + 1)
            sing[n_sing++] = all[i];
        i = j;
    }
    size_t n_out = 0;
    for (u32 i = 0; i < n_ctg; ++i) /* `all` is reused as the output list */
        n_out += agco_find_splitters_in_contig(d + off[i], off[i + 1] - off[i], k, segment_size, sing, n_sing, all + n_out);
    n_out = sort_unique(all, n_out);
    *h_n_spl = n_out;
    int rc = AGC_HIP_OK;
    if (n_out > cap)
        rc = fail(c, AGC_HIP_ECAP, "splitter buffer");
    else if (n_out)
        memcpy(h_spl, all, n_out * sizeof(u64));
    free(all);
    free(sing);
    return rc;
}

int agc_hip_scan_contigs_dev(agc_hip_ctx *c, const uint8_t *d, const uint64_t *off, uint32_t n_ctg, uint32_t k, uint64_t cap,
                             uint64_t *h_n_hits, uint32_t *h_ctg, uint64_t *h_pos, uint64_t *h_dir, uint64_t *h_rc)
{
    if (!c || !h_n_hits || (n_ctg && (!d || !off)))
        return AGC_HIP_EINVAL;
    u64 n_hits = 0;
    for (u32 i = 0; i < n_ctg; ++i) {
        const u64 n = off[i + 1] - off[i];
        size_t scap = n / (k ? k : 1) + 4;
        u64 *buf = (u64 *)malloc(scap * 6 * sizeof(u64));
        u8 *fl = (u8 *)malloc(scap * 2);
        u64 *ss = buf, *sl = buf + scap, *fd = buf + 2 * scap, *fr = buf + 3 * scap, *bd = buf + 4 * scap, *br = buf + 5 * scap;
        const size_t ns = agco_scan_contig(d + off[i], n, k, c->spl, c->n_spl, scap, ss, sl, fd, fr, fl, bd, br, fl + scap);
        for (size_t s = 0; s < ns && s < scap; ++s)
            if (fl[scap + s]) { /* the segment ends with a splitter: one accepted hit at its last symbol */
                if (n_hits < cap) {
                    h_ctg[n_hits] = i;
                    h_pos[n_hits] = ss[s] + sl[s] - 1;
                    h_dir[n_hits] = bd[s];
                    h_rc[n_hits] = br[s];
                }
                ++n_hits;
            }
        free(buf);
        free(fl);
    }
    *h_n_hits = n_hits;
    return n_hits > cap ? fail(c, AGC_HIP_ECAP, "hit buffer") : AGC_HIP_OK;
}

static void *find_ref(agc_hip_ctx *c, u32 gid) { return gid < c->n_lz ? c->lz[gid] : NULL; }

int agc_hip_ref_register_batch_dev(agc_hip_ctx *c, uint32_t n, const uint32_t *gid, const uint8_t *d, const uint64_t *off,
                                   const uint32_t *len, const uint8_t *rc, uint32_t mml)
{
    if (!c || (n && (!gid || !d || !off || !len)))
        return AGC_HIP_EINVAL;
    for (u32 i = 0; i < n; ++i) {
        if (gid[i] >= c->n_lz) {
            u32 m = gid[i] + gid[i] / 2 + 64;
            c->lz = (void **)realloc(c->lz, m * sizeof(void *));
            memset(c->lz + c->n_lz, 0, (m - c->n_lz) * sizeof(void *));
            c->n_lz = m;
        }
        if (c->lz[gid[i]])
            return fail(c, AGC_HIP_EINVAL, "group registered twice");
        u8 *t = slice(d, off[i], len[i], rc && rc[i]);
        c->lz[gid[i]] = agco_lz_create(t, len[i], mml);
        free(t);
    }
    return AGC_HIP_OK;
}

int agc_hip_ref_register(agc_hip_ctx *c, uint32_t gid, const uint8_t *h_ref, uint32_t n, uint32_t mml)
{
    const uint64_t off = 0;
    return agc_hip_ref_register_batch_dev(c, 1, &gid, h_ref, &off, &n, NULL, mml);
}

int agc_hip_lz_encode_batch_dev(agc_hip_ctx *c, uint32_t n, const uint32_t *gid, const uint8_t *d, const uint64_t *off,
                                const uint32_t *len, const uint8_t *rc, uint8_t *h_enc, uint64_t cap, uint64_t *h_enc_off)
{
    if (!c || !h_enc_off || (n && (!gid || !d || !off || !len)))
        return AGC_HIP_EINVAL;
    h_enc_off[0] = 0;
    int over = 0;
    for (u32 i = 0; i < n; ++i) {
        void *z = find_ref(c, gid[i]);
        if (!z)
            return fail(c, AGC_HIP_ENOREF, "encode: unknown group");
        u8 *t = slice(d, off[i], len[i], rc && rc[i]);
        u8 *o = (u8 *)malloc((size_t)len[i] * 2 + 128);
        const size_t m = agco_lz_encode(z, t, len[i], o);
        if (!over && h_enc_off[i] + m <= cap)
            memcpy(h_enc + h_enc_off[i], o, m);
        else
            over = 1;
        h_enc_off[i + 1] = h_enc_off[i] + m;
        free(o);
        free(t);
    }
    return over ? fail(c, AGC_HIP_ECAP, "encode buffer") : AGC_HIP_OK;
}

static void *dup_mem(const void *p, size_t n)
{
    void *q = malloc(n ? n : 1);
    if (p && n)
        memcpy(q, p, n);
    return q;
}

static void enc_clear(agc_hip_ctx *c, uint32_t lane)
{
    free(c->enc[lane].out);
    free(c->enc[lane].eoff);
    c->enc[lane].out = NULL;
    c->enc[lane].eoff = NULL;
    c->enc[lane].pending = 0;
}

/* The device runs the parse in stream order behind `begin` and makes whatever overwrites the sample buffer wait for it (api.hip,
 * Lane2::done); the stand-in has no streams, so it models the same thing by parsing AT begin into a buffer of its own.  `_end`
 * (possibly on another thread: it only touches the enc_* fields) hands the result out. */
static int encode_begin_on(agc_hip_ctx *c, uint32_t lane, uint32_t n, const uint32_t *gid, const uint8_t *d, const uint64_t *off,
                           const uint32_t *len, const uint8_t *rc)
{
    if (!c || lane >= AGC_HIP_ENCODE_LANES || (n && (!gid || !d || !off || !len)))
        return AGC_HIP_EINVAL;
    if (c->enc[lane].pending)
        enc_clear(c, lane); /* an abandoned encode is dropped */
    c->enc[lane].n = n;
    c->enc[lane].eoff = (u64 *)calloc((size_t)n + 1, 8);
    u64 cap = 1u << 16;
    for (u32 i = 0; i < n; ++i)
        cap += len[i] / 64 + 16;
    for (;;) {
        free(c->enc[lane].out);
        c->enc[lane].out = (u8 *)malloc(cap ? cap : 1);
        if (!c->enc[lane].out || !c->enc[lane].eoff)
            return AGC_HIP_ENOMEM;
        const int r = agc_hip_lz_encode_batch_dev(c, n, gid, d, off, len, rc, c->enc[lane].out, cap, c->enc[lane].eoff);
        if (r == AGC_HIP_ECAP) {
            cap = c->enc[lane].eoff[n] + 64;
            continue;
        }
        if (r != AGC_HIP_OK) {
            enc_clear(c, lane);
            return r;
        }
        break;
    }
    c->enc[lane].pending = 1;
    return AGC_HIP_OK;
}

int agc_hip_lz_encode_begin_dev(agc_hip_ctx *c, uint32_t n, const uint32_t *gid, const uint8_t *d, const uint64_t *off, const uint32_t *len,
                                const uint8_t *rc)
{
    return encode_begin_on(c, 0, n, gid, d, off, len, rc);
}

int agc_hip_lz_encode_pending(agc_hip_ctx *c, uint32_t *h_n) { return agc_hip_lz_encode_pending_on(c, 0, h_n); }
int agc_hip_lz_encode_end(agc_hip_ctx *c, uint8_t *h_enc, uint64_t cap, uint64_t *h_enc_off)
{
    return agc_hip_lz_encode_end_on(c, 0, h_enc, cap, h_enc_off);
}

int agc_hip_lz_encode_pending_on(agc_hip_ctx *c, uint32_t lane, uint32_t *h_n)
{
    if (!c || !h_n || lane >= AGC_HIP_ENCODE_LANES)
        return AGC_HIP_EINVAL;
    *h_n = c->enc[lane].pending ? c->enc[lane].n : 0;
    return AGC_HIP_OK;
}

int agc_hip_lz_encode_drop_on(agc_hip_ctx *c, uint32_t lane)
{
    if (!c || lane >= AGC_HIP_ENCODE_LANES)
        return AGC_HIP_EINVAL;
    if (c->enc[lane].pending)
        enc_clear(c, lane);
    return AGC_HIP_OK;
}

int agc_hip_lz_encode_end_on(agc_hip_ctx *c, uint32_t lane, uint8_t *h_enc, uint64_t cap, uint64_t *h_enc_off)
{
    if (!c || !h_enc_off || lane >= AGC_HIP_ENCODE_LANES)
        return AGC_HIP_EINVAL;
    if (!c->enc[lane].pending)
        return fail(c, AGC_HIP_EINVAL, "encode_end: no encode in flight");
    const u32 n = c->enc[lane].n;
    memcpy(h_enc_off, c->enc[lane].eoff, ((size_t)n + 1) * 8);
    if (c->enc[lane].eoff[n] > cap)
        return AGC_HIP_ECAP; /* (still in flight: call again with a larger buffer) */
    if (c->enc[lane].eoff[n]) {
        if (!h_enc)
            return AGC_HIP_EINVAL;
        memcpy(h_enc, c->enc[lane].out, c->enc[lane].eoff[n]);
    }
    enc_clear(c, lane);
    return AGC_HIP_OK;
}

int agc_hip_zstd17_background(agc_hip_ctx *c, int on)
{
    (void)on;
    return c ? AGC_HIP_OK : AGC_HIP_EINVAL;
}

int agc_hip_host_alloc(agc_hip_ctx *c, uint64_t bytes, void **out)
{
    if (!c || !out)
        return AGC_HIP_EINVAL;
    *out = malloc(bytes ? bytes : 1);
    return *out ? AGC_HIP_OK : AGC_HIP_ENOMEM;
}

int agc_hip_host_free(agc_hip_ctx *c, void *p)
{
    if (!c)
        return AGC_HIP_EINVAL;
    free(p);
    return AGC_HIP_OK;
}

int agc_hip_lz_estimate_batch_dev(agc_hip_ctx *c, uint32_t n, const uint32_t *gid, const uint8_t *d, const uint64_t *off,
                                  const uint32_t *len, const uint8_t *rc, uint32_t *h_cost, uint32_t *h_peak)
{
    if (!c || (n && (!gid || !d || !off || !len || !h_cost)))
        return AGC_HIP_EINVAL;
    for (u32 i = 0; i < n; ++i) {
        void *z = find_ref(c, gid[i]);
        if (!z)
            return fail(c, AGC_HIP_ENOREF, "estimate: unknown group");
        u8 *t = slice(d, off[i], len[i], rc && rc[i]);
        u32 peak = 0;
        h_cost[i] = agco_lz_estimate(z, t, len[i], 0xFFFFFFFFu, &peak);
        if (h_peak)
            h_peak[i] = peak;
        free(t);
    }
    return AGC_HIP_OK;
}

/* src/core/agc_compressor.cpp:1538-1617 */
int agc_hip_lz_split_point_batch_dev(agc_hip_ctx *c, uint32_t n, const uint32_t *g1, const uint32_t *g2, const uint8_t *d,
                                     const uint64_t *off, const uint32_t *len, const uint8_t *rc1, const uint8_t *pf1, const uint8_t *rc2,
                                     const uint8_t *pf2, uint32_t *best_pos, uint32_t *best_sum)
{
    if (!c || (n && (!g1 || !g2 || !d || !off || !len || !rc1 || !pf1 || !rc2 || !pf2 || !best_pos)))
        return AGC_HIP_EINVAL;
    for (u32 s = 0; s < n; ++s) {
        void *z1 = find_ref(c, g1[s]), *z2 = find_ref(c, g2[s]);
        if (!z1 || !z2)
            return fail(c, AGC_HIP_ENOREF, "split point: unknown group");
        const u32 m = len[s];
        u32 *v1 = (u32 *)calloc((size_t)m + 1, 4), *v2 = (u32 *)calloc((size_t)m + 1, 4);
        u8 *t = slice(d, off[s], m, rc1[s]);
        agco_lz_cost_vector(z1, t, m, pf1[s], v1);
        free(t);
        if (!pf1[s])
            for (u32 i = 0; i < m / 2; ++i) {
                u32 x = v1[i];
                v1[i] = v1[m - 1 - i];
                v1[m - 1 - i] = x;
            }
        for (u32 i = 1; i < m; ++i)
            v1[i] += v1[i - 1];
        t = slice(d, off[s], m, rc2[s]);
        agco_lz_cost_vector(z2, t, m, pf2[s], v2);
        free(t);
        if (!pf2[s]) { /* suffix sums in place */
            for (u32 i = m; i-- > 1;)
                v2[i - 1] += v2[i];
        } else { /* prefix sums, then reversed */
            for (u32 i = 1; i < m; ++i)
                v2[i] += v2[i - 1];
            for (u32 i = 0; i < m / 2; ++i) {
                u32 x = v2[i];
                v2[i] = v2[m - 1 - i];
                v2[m - 1 - i] = x;
            }
        }
        u32 bs = ~0u, bp = 0;
        for (u32 i = 0; i < m; ++i) {
            const u32 cs = v1[i] + v2[i];
            if (cs < bs) {
                bs = cs;
                bp = i;
            }
        }
        best_pos[s] = bp;
        if (best_sum)
            best_sum[s] = bs;
        free(v1);
        free(v2);
    }
    return AGC_HIP_OK;
}

int agc_hip_fetch_slices_dev(agc_hip_ctx *c, uint32_t n, const uint8_t *d, const uint64_t *off, const uint32_t *len, const uint8_t *rc,
                             uint8_t *h_out, uint64_t cap, uint64_t *h_out_off)
{
    if (!c || !h_out_off || (n && (!d || !off || !len)))
        return AGC_HIP_EINVAL;
    h_out_off[0] = 0;
    for (u32 i = 0; i < n; ++i)
        h_out_off[i + 1] = h_out_off[i] + len[i];
    if (h_out_off[n] > cap)
        return fail(c, AGC_HIP_ECAP, "fetch buffer");
    for (u32 i = 0; i < n; ++i) {
        if (rc && rc[i])
            agco_rev_comp(d + off[i], len[i], h_out + h_out_off[i]);
        else
            memcpy(h_out + h_out_off[i], d + off[i], len[i]);
    }
    return AGC_HIP_OK;
}

int agc_hip_ref_lag_counts_dev(agc_hip_ctx *c, uint32_t n, const uint8_t *d, const uint64_t *off, const uint32_t *len, const uint8_t *rc,
                               uint32_t *h_cnt, uint32_t *h_cur)
{
    if (!c || (n && (!d || !off || !len || !h_cnt || !h_cur)))
        return AGC_HIP_EINVAL;
    for (u32 i = 0; i < n; ++i) {
        u8 *t = slice(d, off[i], len[i], rc && rc[i]);
        agco_ref_lag_counts(t, len[i], h_cnt + 28 * i, h_cur + 28 * i);
        free(t);
    }
    return AGC_HIP_OK;
}

/* ---- 2-bit packed samples: the layout of include/agc_hip.h on the host ---- */
uint64_t agc_hip_packed_words_bytes(uint64_t n) { return ((n + 1023) / 1024) * 256 + 64; }
uint64_t agc_hip_packed_index_bytes(uint64_t n) { return ((n + 1023) / 1024) * 4 + 64; }

int agc_hip_pack_dev(agc_hip_ctx *c, const uint8_t *codes, uint64_t n, uint32_t *words, int32_t *esc_index, uint8_t *esc_bytes, uint64_t cap,
                     uint64_t *h_n_esc)
{
    if (!c || !h_n_esc)
        return AGC_HIP_EINVAL;
    uint64_t cnt = 0;
    for (uint64_t b = 0; b * 1024 < n; ++b) {
        int high = 0;
        for (uint64_t i = b * 1024; i < n && i < (b + 1) * 1024; ++i)
            high |= codes[i] > 3;
        for (uint32_t w = 0; w < 64; ++w) {
            uint32_t x = 0;
            for (uint32_t j = 0; j < 16; ++j) {
                const uint64_t i = b * 1024 + w * 16 + j;
                x |= (uint32_t)((i < n ? codes[i] : 0) & 3) << (2 * j);
            }
            words[b * 64 + w] = x;
        }
        esc_index[b] = -1;
        if (high) {
            if (cnt < cap) {
                esc_index[b] = (int32_t)cnt;
                for (uint32_t j = 0; j < 1024; ++j)
                    esc_bytes[cnt * 1024 + j] = b * 1024 + j < n ? codes[b * 1024 + j] : 0;
            }
            ++cnt;
        }
    }
    *h_n_esc = cnt;
    return cnt > cap ? AGC_HIP_ECAP : AGC_HIP_OK;
}

int agc_hip_expand_dev(agc_hip_ctx *c, const agc_hip_packed *pk, uint8_t *codes)
{
    if (!c || !pk)
        return AGC_HIP_EINVAL;
    for (uint64_t i = 0; i < pk->n_symbols; ++i) {
        const int32_t s = pk->d_esc_index[i / 1024];
        codes[i] = s >= 0 ? pk->d_esc_bytes[(uint64_t)s * 1024 + (i & 1023)] : (uint8_t)((pk->d_words[i >> 4] >> (2 * (i & 15))) & 3);
    }
    return AGC_HIP_OK;
}

int agc_hip_scan_packed_dev(agc_hip_ctx *c, const agc_hip_packed *pk, const uint64_t *h_ctg_off, uint32_t n_ctg, uint32_t k, uint64_t cap,
                            uint64_t *h_n_hits, uint32_t *h_hit_ctg, uint64_t *h_hit_pos, uint64_t *h_hit_dir, uint64_t *h_hit_rc)
{
    if (!c || !pk || k < 16)
        return AGC_HIP_EINVAL;
    uint8_t *codes = (uint8_t *)malloc(pk->n_symbols + 64);
    agc_hip_expand_dev(c, pk, codes);
    const int r = agc_hip_scan_contigs_dev(c, codes, h_ctg_off, n_ctg, k, cap, h_n_hits, h_hit_ctg, h_hit_pos, h_hit_dir, h_hit_rc);
    free(codes);
    return r;
}

/* the prefetch entry points: the stand-in remembers which sample was announced and scans when asked (same results, no concurrency) */
int agc_hip_prefetch_packed_dev(agc_hip_ctx *c, const agc_hip_packed *pk, const uint64_t *h_ctg_off, uint32_t n_ctg, uint32_t k)
{
    if (!c || !pk || !h_ctg_off || !n_ctg || k < 16 || !pk->n_symbols)
        return AGC_HIP_EINVAL;
    c->pf_words = pk->d_words;
    return AGC_HIP_OK;
}
int agc_hip_scan_prefetched(agc_hip_ctx *c, const agc_hip_packed *pk, const uint64_t *h_ctg_off, uint32_t n_ctg, uint32_t k, uint64_t cap,
                            uint64_t *h_n_hits, uint32_t *h_hit_ctg, uint64_t *h_hit_pos, uint64_t *h_hit_dir, uint64_t *h_hit_rc)
{
    if (!c || !pk || c->pf_words != pk->d_words)
        return AGC_HIP_EINVAL;
    return agc_hip_scan_packed_dev(c, pk, h_ctg_off, n_ctg, k, cap, h_n_hits, h_hit_ctg, h_hit_pos, h_hit_dir, h_hit_rc);
}

/* codes -> the context's own packed buffers */
int agc_hip_sample_pack(agc_hip_ctx *c, const uint8_t *d_codes, uint64_t n, agc_hip_packed *out)
{
    if (!c || !out || (n && !d_codes))
        return AGC_HIP_EINVAL;
    const uint64_t nb = (n + 1023) / 1024;
    free(c->sp_words);
    free(c->sp_index);
    free(c->sp_esc);
    c->sp_words = (uint32_t *)malloc(agc_hip_packed_words_bytes(n));
    c->sp_index = (int32_t *)malloc(agc_hip_packed_index_bytes(n));
    c->sp_esc = (u8 *)malloc(nb * 1024 + 64);
    uint64_t cnt = 0;
    const int r = agc_hip_pack_dev(c, d_codes, n, c->sp_words, c->sp_index, c->sp_esc, nb, &cnt);
    out->d_words = c->sp_words;
    out->d_esc_index = c->sp_index;
    out->d_esc_bytes = c->sp_esc;
    out->n_symbols = n;
    return r;
}

/* The *_packed entry points: the stand-in expands the range the sequences span and calls the byte variant (the device reads the
 * packed words in place, sym_view.h; the results are the same by the tests of both). */
static u8 *expand_span(const agc_hip_packed *pk, u32 n, const u64 *off, const u32 *len, u64 **off2)
{
    u64 lo = ~0ull, hi = 0;
    for (u32 i = 0; i < n; ++i)
        if (len[i]) {
            if (off[i] < lo)
                lo = off[i];
            if (off[i] + len[i] > hi)
                hi = off[i] + len[i];
        }
    *off2 = (u64 *)calloc((size_t)n + 1, 8);
    if (hi <= lo)
        return (u8 *)calloc(64, 1);
    if (hi > pk->n_symbols)
        return NULL;
    u8 *buf = (u8 *)malloc(hi - lo + 64);
    for (u64 i = lo; i < hi; ++i) {
        const int32_t s_ = pk->d_esc_index[i / 1024];
        buf[i - lo] = s_ >= 0 ? pk->d_esc_bytes[(u64)s_ * 1024 + (i & 1023)] : (u8)((pk->d_words[i >> 4] >> (2 * (i & 15))) & 3);
    }
    for (u32 i = 0; i < n; ++i)
        (*off2)[i] = len[i] ? off[i] - lo : 0;
    return buf;
}
#define WITH_SPAN(call)                                             \
    u64 *off2 = NULL;                                               \
    u8 *buf = expand_span(pk, n, off, len, &off2);                  \
    if (!buf) {                                                     \
        free(off2);                                                 \
        return fail(c, AGC_HIP_EINVAL, "sequence past the buffer"); \
    }                                                               \
    const int r_ = (call);                                          \
    free(buf);                                                      \
    free(off2);                                                     \
    return r_;

int agc_hip_ref_register_batch_packed(agc_hip_ctx *c, uint32_t n, const uint32_t *gid, const agc_hip_packed *pk, const uint64_t *off,
                                      const uint32_t *len, const uint8_t *rc, uint32_t mml)
{
    if (!c || (n && (!gid || !pk || !off || !len)))
        return AGC_HIP_EINVAL;
    WITH_SPAN(agc_hip_ref_register_batch_dev(c, n, gid, buf, off2, len, rc, mml))
}
int agc_hip_lz_encode_batch_packed(agc_hip_ctx *c, uint32_t n, const uint32_t *gid, const agc_hip_packed *pk, const uint64_t *off,
                                   const uint32_t *len, const uint8_t *rc, uint8_t *h_enc, uint64_t cap, uint64_t *h_enc_off)
{
    if (!c || !h_enc_off || (n && (!gid || !pk || !off || !len)))
        return AGC_HIP_EINVAL;
    WITH_SPAN(agc_hip_lz_encode_batch_dev(c, n, gid, buf, off2, len, rc, h_enc, cap, h_enc_off))
}
int agc_hip_lz_encode_begin_packed_on(agc_hip_ctx *c, uint32_t lane, uint32_t n, const uint32_t *gid, const agc_hip_packed *pk, const uint64_t *off,
                                      const uint32_t *len, const uint8_t *rc)
{
    if (!c || lane >= AGC_HIP_ENCODE_LANES || (n && (!gid || !pk || !off || !len)))
        return AGC_HIP_EINVAL;
    WITH_SPAN(encode_begin_on(c, lane, n, gid, buf, off2, len, rc))
}
int agc_hip_lz_encode_begin_packed(agc_hip_ctx *c, uint32_t n, const uint32_t *gid, const agc_hip_packed *pk, const uint64_t *off,
                                   const uint32_t *len, const uint8_t *rc)
{
    return agc_hip_lz_encode_begin_packed_on(c, 0, n, gid, pk, off, len, rc);
}
int agc_hip_lz_estimate_batch_packed(agc_hip_ctx *c, uint32_t n, const uint32_t *gid, const agc_hip_packed *pk, const uint64_t *off,
                                     const uint32_t *len, const uint8_t *rc, uint32_t *h_cost, uint32_t *h_peak)
{
    if (!c || (n && (!gid || !pk || !off || !len || !h_cost)))
        return AGC_HIP_EINVAL;
    WITH_SPAN(agc_hip_lz_estimate_batch_dev(c, n, gid, buf, off2, len, rc, h_cost, h_peak))
}
int agc_hip_lz_cost_vector_batch_dev(agc_hip_ctx *c, uint32_t n, const uint32_t *gid, const uint8_t *d, const uint64_t *off,
                                     const uint32_t *len, const uint8_t *rc, const uint8_t *pf, uint32_t *h_costs)
{
    if (!c || (n && (!gid || !d || !off || !len || !h_costs)))
        return AGC_HIP_EINVAL;
    u64 o = 0;
    for (u32 i = 0; i < n; ++i) {
        void *z = find_ref(c, gid[i]);
        if (!z)
            return fail(c, AGC_HIP_ENOREF, "cost vector: unknown group");
        u8 *t = slice(d, off[i], len[i], rc && rc[i]);
        agco_lz_cost_vector(z, t, len[i], pf && pf[i], h_costs + o);
        o += len[i];
        free(t);
    }
    return AGC_HIP_OK;
}
int agc_hip_lz_cost_vector_batch_packed(agc_hip_ctx *c, uint32_t n, const uint32_t *gid, const agc_hip_packed *pk, const uint64_t *off,
                                        const uint32_t *len, const uint8_t *rc, const uint8_t *pf, uint32_t *h_costs)
{
    if (!c || (n && (!gid || !pk || !off || !len || !h_costs)))
        return AGC_HIP_EINVAL;
    WITH_SPAN(agc_hip_lz_cost_vector_batch_dev(c, n, gid, buf, off2, len, rc, pf, h_costs))
}
int agc_hip_lz_split_point_batch_packed(agc_hip_ctx *c, uint32_t n, const uint32_t *g1, const uint32_t *g2, const agc_hip_packed *pk,
                                        const uint64_t *off, const uint32_t *len, const uint8_t *rc1, const uint8_t *pf1, const uint8_t *rc2,
                                        const uint8_t *pf2, uint32_t *best_pos, uint32_t *best_sum)
{
    if (!c || (n && (!g1 || !g2 || !pk || !off || !len)))
        return AGC_HIP_EINVAL;
    WITH_SPAN(agc_hip_lz_split_point_batch_dev(c, n, g1, g2, buf, off2, len, rc1, pf1, rc2, pf2, best_pos, best_sum))
}
int agc_hip_fetch_slices_packed(agc_hip_ctx *c, uint32_t n, const agc_hip_packed *pk, const uint64_t *off, const uint32_t *len, const uint8_t *rc,
                                uint8_t *h_out, uint64_t cap, uint64_t *h_out_off)
{
    if (!c || !h_out_off || (n && (!pk || !off || !len)))
        return AGC_HIP_EINVAL;
    WITH_SPAN(agc_hip_fetch_slices_dev(c, n, buf, off2, len, rc, h_out, cap, h_out_off))
}
int agc_hip_ref_lag_counts_packed(agc_hip_ctx *c, uint32_t n, const agc_hip_packed *pk, const uint64_t *off, const uint32_t *len, const uint8_t *rc,
                                  uint32_t *h_cnt, uint32_t *h_cur)
{
    if (!c || (n && (!pk || !off || !len || !h_cnt || !h_cur)))
        return AGC_HIP_EINVAL;
    WITH_SPAN(agc_hip_ref_lag_counts_dev(c, n, buf, off2, len, rc, h_cnt, h_cur))
}

/* ---- segments and their groups (include/agc_hip.h): the oracle's compress_contig per contig, a probe of the mirrored table, and
 * -- encode_known -- the encode of the segments whose group is known, parsed at once like agc_hip_lz_encode_begin_dev ---- */
uint64_t agc_hip_group_hash(uint64_t k1, uint64_t k2)
{
    uint64_t h = k1 * 0x9E3779B97F4A7C15ULL;
    h ^= (h >> 32) ^ (k2 * 0xC2B2AE3D27D4EB4FULL);
    return h ^ (h >> 29);
}

int agc_hip_group_map_set(agc_hip_ctx *c, const agc_hip_group_slot *h_slots, uint64_t n_slots)
{
    if (!c || (n_slots && !h_slots) || (n_slots & (n_slots - 1)))
        return AGC_HIP_EINVAL;
    free(c->gmap);
    c->gmap = (agc_hip_group_slot *)malloc((n_slots ? n_slots : 1) * sizeof(agc_hip_group_slot));
    if (n_slots)
        memcpy(c->gmap, h_slots, n_slots * sizeof(agc_hip_group_slot));
    c->gmap_slots = n_slots;
    return AGC_HIP_OK;
}

int agc_hip_group_map_update(agc_hip_ctx *c, uint32_t n, const uint64_t *h_idx, const agc_hip_group_slot *h_slots)
{
    if (!c || (n && (!h_idx || !h_slots || !c->gmap_slots)))
        return AGC_HIP_EINVAL;
    for (u32 i = 0; i < n; ++i) {
        if (h_idx[i] >= c->gmap_slots)
            return AGC_HIP_EINVAL;
        c->gmap[h_idx[i]] = h_slots[i];
    }
    return AGC_HIP_OK;
}

int agc_hip_segments_packed(agc_hip_ctx *c, const agc_hip_packed *pk, const uint64_t *h_ctg_off, uint32_t n_ctg, uint32_t k, int prefetched,
                            int encode_known, uint64_t cap, agc_hip_segment *h_segs, uint64_t *h_n_segs, uint32_t *h_n_encoded)
{
    if (!c || !pk || !h_ctg_off || !h_n_segs || k < 16 || k > 32)
        return AGC_HIP_EINVAL;
    if (prefetched && c->pf_words != pk->d_words)
        return AGC_HIP_EINVAL;
    if (encode_known && c->enc[0].pending)
        return fail(c, AGC_HIP_EINVAL, "segments_packed: the previous encode was not collected");
    *h_n_segs = 0;
    if (h_n_encoded)
        *h_n_encoded = 0;
    if (!n_ctg)
        return AGC_HIP_OK;
    u8 *codes = (u8 *)malloc(pk->n_symbols + 64);
    agc_hip_expand_dev(c, pk, codes);
    u64 n = 0;
    int over = 0;
    for (u32 ci = 0; ci < n_ctg; ++ci) {
        const u64 len = h_ctg_off[ci + 1] - h_ctg_off[ci];
        u64 scap = len / (k ? k : 1) + 2;
        u64 *st_ = (u64 *)malloc(scap * 8 * 6);
        u8 *fl = (u8 *)malloc(scap * 2);
        const size_t m = agco_scan_contig(codes + h_ctg_off[ci], len, k, c->spl, c->n_spl, scap, st_, st_ + scap, st_ + 2 * scap, st_ + 3 * scap, fl,
                                          st_ + 4 * scap, st_ + 5 * scap, fl + scap);
        for (size_t i = 0; i < m; ++i, ++n) {
            if (n >= cap) {
                over = 1;
                continue;
            }
            agc_hip_segment *s_ = &h_segs[n];
            memset(s_, 0, sizeof *s_);
            s_->ctg = ci;
            s_->start = st_[i];
            s_->len = (u32)st_[scap + i];
            s_->front_full = fl[i];
            s_->back_full = fl[scap + i];
            if (s_->front_full) {
                s_->front_dir = st_[2 * scap + i];
                s_->front_rc = st_[3 * scap + i];
            }
            if (s_->back_full) {
                s_->back_dir = st_[4 * scap + i];
                s_->back_rc = st_[5 * scap + i];
            }
            s_->map_gid = -1;
            if (s_->front_full && s_->back_full) {
                const u64 f = s_->front_dir < s_->front_rc ? s_->front_dir : s_->front_rc, b = s_->back_dir < s_->back_rc ? s_->back_dir : s_->back_rc;
                const u64 k1 = f < b ? f : b, k2 = f < b ? b : f;
                s_->store_rc = f < b ? 0 : 1;
                if (c->gmap_slots)
                    for (u64 j = agc_hip_group_hash(k1, k2) & (c->gmap_slots - 1); c->gmap[j].used; j = (j + 1) & (c->gmap_slots - 1))
                        if (c->gmap[j].k1 == k1 && c->gmap[j].k2 == k2) {
                            s_->map_gid = c->gmap[j].gid;
                            break;
                        }
            }
        }
        free(st_);
        free(fl);
    }
    if (over) {
        free(codes);
        *h_n_segs = n;
        return AGC_HIP_ECAP;
    }
    *h_n_segs = n;
    int r = AGC_HIP_OK;
    free(c->sg_gid), free(c->sg_len), free(c->sg_off), free(c->sg_rc), free(c->sg_codes);
    c->sg_gid = c->sg_len = NULL;
    c->sg_off = NULL;
    c->sg_rc = c->sg_codes = NULL;
    c->sg_valid = 0;
    if (c->gmap_slots) {
        u32 ne = 0;
        u32 *gid = (u32 *)malloc((n + 1) * 4), *len = (u32 *)malloc((n + 1) * 4);
        u64 *off = (u64 *)malloc((n + 1) * 8);
        u8 *rc = (u8 *)malloc(n + 1);
        for (u64 i = 0; i < n; ++i) {
            agc_hip_segment *s_ = &h_segs[i];
            if (s_->front_full && s_->back_full && s_->map_gid >= 16 && find_ref(c, (u32)s_->map_gid)) {
                if (encode_known)
                    s_->encoded = 1;
                gid[ne] = (u32)s_->map_gid;
                off[ne] = h_ctg_off[s_->ctg] + s_->start;
                len[ne] = s_->len;
                rc[ne] = s_->store_rc;
                ++ne;
            }
        }
        if (encode_known) {
            if (ne) {
                r = agc_hip_lz_encode_begin_dev(c, ne, gid, codes, off, len, rc);
                if (r == AGC_HIP_OK && h_n_encoded)
                    *h_n_encoded = ne;
            }
            free(gid), free(len), free(off), free(rc);
        } else { /* kept for agc_hip_segments_encode_known */
            c->sg_gid = gid, c->sg_len = len, c->sg_off = off, c->sg_rc = rc, c->sg_codes = codes, c->sg_ne = ne, c->sg_valid = 1;
            codes = NULL;
        }
    }
    free(codes);
    return r;
}

/* the launch as a call of its own: the stand-in kept what it needs from the last agc_hip_segments_packed call */
int agc_hip_segments_encode_known(agc_hip_ctx *c)
{
    const uint32_t lane = 0;
    if (!c || !c->sg_valid)
        return fail(c, AGC_HIP_EINVAL, "segments_encode_known: no segments");
    if (c->enc[lane].pending)
        return fail(c, AGC_HIP_EINVAL, "segments_encode_known: the previous encode was not collected");
    c->sg_valid = 0;
    if (!c->sg_ne) {
        /* (nothing known: an empty encode is in flight, as on the device) */
        c->enc[lane].n = 0;
        c->enc[lane].eoff = (u64 *)calloc(1, 8);
        c->enc[lane].out = (u8 *)malloc(1);
        c->enc[lane].pending = 1;
        return AGC_HIP_OK;
    }
    return agc_hip_lz_encode_begin_dev(c, c->sg_ne, c->sg_gid, c->sg_codes, c->sg_off, c->sg_len, c->sg_rc);
}

/* a1 on the stand-in: the oracle's preprocess_raw_contig */
size_t agco_preprocess(const u8 *raw, size_t n, u8 *out);
int agc_hip_preprocess_dev(agc_hip_ctx *c, const uint8_t *d_raw, uint64_t n_raw, uint8_t *d_codes, uint64_t *h_n)
{
    if (!c || !h_n)
        return AGC_HIP_EINVAL;
    *h_n = n_raw ? agco_preprocess(d_raw, n_raw, d_codes) : 0;
    return AGC_HIP_OK;
}
int agc_hip_preprocess(agc_hip_ctx *c, const uint8_t *h_raw, uint64_t n_raw, uint8_t *d_codes, uint64_t *h_n)
{
    return agc_hip_preprocess_dev(c, h_raw, n_raw, d_codes, h_n);
}

/* S1a on the stand-in: the oracle's preprocess_raw_contig of every contig's byte range, the codes back to back, then the
 * stand-in's own pack.  "begin" does the work, "end" hands the result out (one conversion in flight, as in the product). */
int agc_hip_pack_fasta_begin(agc_hip_ctx *c, const uint8_t *d_raw, uint64_t n_raw, const uint64_t *h_raw_begin, const uint64_t *h_raw_end, uint32_t n_ctg,
                             uint32_t *d_words, int32_t *d_esc_index, uint8_t *d_esc_bytes, uint64_t esc_cap_blocks)
{
    if (!c || (n_ctg && (!h_raw_begin || !h_raw_end)))
        return AGC_HIP_EINVAL;
    u64 ub = 0, prev = 0;
    for (u32 i = 0; i < n_ctg; ++i) {
        if (h_raw_begin[i] < prev || h_raw_end[i] < h_raw_begin[i] || h_raw_end[i] > n_raw)
            return AGC_HIP_EINVAL;
        prev = h_raw_end[i];
        ub += h_raw_end[i] - h_raw_begin[i];
    }
    u8 *codes = (u8 *)malloc(ub + 1);
    free(c->pfa_off);
    c->pfa_off = (u64 *)calloc((size_t)n_ctg + 1, 8);
    if (!codes || !c->pfa_off) {
        free(codes);
        return fail(c, AGC_HIP_ENOMEM, "out of memory (pack_fasta)");
    }
    u64 n = 0;
    for (u32 i = 0; i < n_ctg; ++i) {
        c->pfa_off[i] = n;
        if (h_raw_end[i] > h_raw_begin[i])
            n += agco_preprocess(d_raw + h_raw_begin[i], (size_t)(h_raw_end[i] - h_raw_begin[i]), codes + n);
    }
    c->pfa_off[n_ctg] = n;
    c->pfa_n_ctg = n_ctg;
    c->pfa_esc = 0;
    c->pfa_rc = n ? agc_hip_pack_dev(c, codes, n, d_words, d_esc_index, d_esc_bytes, esc_cap_blocks, &c->pfa_esc) : AGC_HIP_OK;
    free(codes);
    c->pfa_pending = 1;
    return AGC_HIP_OK;
}

int agc_hip_pack_fasta_end(agc_hip_ctx *c, uint64_t *h_ctg_off, uint64_t *h_n_esc_blocks)
{
    if (!c || !h_ctg_off || !h_n_esc_blocks || !c->pfa_pending)
        return AGC_HIP_EINVAL;
    c->pfa_pending = 0;
    memcpy(h_ctg_off, c->pfa_off, ((size_t)c->pfa_n_ctg + 1) * 8);
    *h_n_esc_blocks = c->pfa_esc;
    return c->pfa_rc;
}

int agc_hip_pack_fasta_dev(agc_hip_ctx *c, const uint8_t *d_raw, uint64_t n_raw, const uint64_t *h_raw_begin, const uint64_t *h_raw_end, uint32_t n_ctg,
                           uint32_t *d_words, int32_t *d_esc_index, uint8_t *d_esc_bytes, uint64_t esc_cap_blocks, uint64_t *h_ctg_off, uint64_t *h_n_esc_blocks)
{
    const int rc = agc_hip_pack_fasta_begin(c, d_raw, n_raw, h_raw_begin, h_raw_end, n_ctg, d_words, d_esc_index, d_esc_bytes, esc_cap_blocks);
    return rc != AGC_HIP_OK ? rc : agc_hip_pack_fasta_end(c, h_ctg_off, h_n_esc_blocks);
}

int agc_hip_sample_pack_fasta(agc_hip_ctx *c, uint32_t n_ctg, const uint8_t *const *h_raw, const uint64_t *h_len, agc_hip_packed *out, uint64_t *h_ctg_off)
{
    if (!c || !out || !h_ctg_off || (n_ctg && (!h_raw || !h_len)))
        return AGC_HIP_EINVAL;
    memset(out, 0, sizeof *out);
    u64 ub = 0;
    for (u32 i = 0; i < n_ctg; ++i)
        ub += h_len[i];
    u8 *codes = (u8 *)malloc(ub + 1);
    if (!codes)
        return fail(c, AGC_HIP_ENOMEM, "out of memory (sample_pack_fasta)");
    u64 n = 0;
    for (u32 i = 0; i < n_ctg; ++i) {
        h_ctg_off[i] = n;
        if (h_len[i])
            n += agco_preprocess(h_raw[i], (size_t)h_len[i], codes + n);
    }
    h_ctg_off[n_ctg] = n;
    const int rc = n ? agc_hip_sample_pack(c, codes, n, out) : AGC_HIP_OK;
    free(codes);
    return rc;
}

/* lag counters + symbols of a registration's new references in one call: done at begin, nothing to wait for at end */
int agc_hip_ref_store_begin_packed(agc_hip_ctx *c, uint32_t slot, uint32_t n_refs, uint32_t n, const agc_hip_packed *pk, const uint64_t *h_off,
                                   const uint32_t *h_len, const uint8_t *h_rc, uint32_t *h_cnt, uint32_t *h_cur, uint8_t *h_out, uint64_t out_cap,
                                   uint64_t *h_out_off)
{
    if (!c || slot >= 2 || n_refs > n || !h_out_off)
        return AGC_HIP_EINVAL;
    if (n_refs) {
        const int r = agc_hip_ref_lag_counts_packed(c, n_refs, pk, h_off, h_len, h_rc, h_cnt, h_cur);
        if (r != AGC_HIP_OK)
            return r;
    }
    h_out_off[0] = 0;
    return n ? agc_hip_fetch_slices_packed(c, n, pk, h_off, h_len, h_rc, h_out, out_cap, h_out_off) : AGC_HIP_OK;
}
int agc_hip_ref_store_end(agc_hip_ctx *c, uint32_t slot) { return c && slot < 2 ? AGC_HIP_OK : AGC_HIP_EINVAL; }
