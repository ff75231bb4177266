// segments.hip -- agc_hip_group_map_* and agc_hip_segments_packed (include/agc_hip.h): the host side of seg_kernels.hip.
// Included by api.hip after splitters.hip (uses its context, its helpers and rocPRIM).

namespace agc {

__global__ void __launch_bounds__(256) gmap_scatter_kernel(GroupSlot *__restrict__ table, const uint64_t *__restrict__ idx, const GroupSlot *__restrict__ slots,
                                                           uint32_t n)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n)
        table[idx[t]] = slots[t];
}

} // namespace agc

extern "C" {

uint64_t agc_hip_group_hash(uint64_t k1, uint64_t k2) { return agc::group_hash(k1, k2); }

int agc_hip_group_map_set(agc_hip_ctx *c, const agc_hip_group_slot *h_slots, uint64_t n_slots)
{
    static_assert(sizeof(agc_hip_group_slot) == sizeof(GroupSlot) && sizeof(GroupSlot) == 24, "slot layout");
    if (!c || (n_slots && !h_slots) || (n_slots & (n_slots - 1)) != 0 || (n_slots && n_slots < 16))
        return AGC_HIP_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    c->gmap_slots = 0;
    if (!n_slots)
        return AGC_HIP_OK;
    CHK(ensure(c, c->d_gmap, n_slots * sizeof(GroupSlot) + 64));
    HIPCHK(c, hipMemcpyAsync(c->d_gmap.p, h_slots, n_slots * sizeof(GroupSlot), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream)); // (the caller's array may change as soon as this returns)
    c->gmap_slots = n_slots;
    return AGC_HIP_OK;
}

int agc_hip_group_map_update(agc_hip_ctx *c, uint32_t n, const uint64_t *h_idx, const agc_hip_group_slot *h_slots)
{
    if (!c || (n && (!h_idx || !h_slots)))
        return AGC_HIP_EINVAL;
    if (!n)
        return AGC_HIP_OK;
    if (!c->gmap_slots)
        return AGC_HIP_EINVAL;
    for (uint32_t i = 0; i < n; ++i)
        if (h_idx[i] >= c->gmap_slots)
            return AGC_HIP_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    // through a pinned staging buffer of the context: nothing waits (the scatter is ordered before every later launch of the
    // context's stream); the buffer is reused once the event behind its last scatter has passed
    const size_t idx_bytes = ((size_t)n * 8 + 15) & ~(size_t)15, need = idx_bytes + (size_t)n * sizeof(GroupSlot);
    if (!c->gmap_ev)
        HIPCHK(c, hipEventCreateWithFlags(&c->gmap_ev, hipEventDisableTiming));
    if (c->gmap_ev_valid)
        HIPCHK(c, hipEventSynchronize(c->gmap_ev));
    if (c->h_gmap_stage_cap < need) {
        if (c->h_gmap_stage)
            HIPCHK(c, hipHostFree(c->h_gmap_stage));
        c->h_gmap_stage = nullptr;
        c->h_gmap_stage_cap = 0;
        HIPCHK(c, hipHostMalloc(&c->h_gmap_stage, need + need / 2 + 4096, hipHostMallocDefault));
        c->h_gmap_stage_cap = need + need / 2 + 4096;
    }
    CHK(ensure(c, c->d_gmap_stage, need + 64));
    memcpy(c->h_gmap_stage, h_idx, (size_t)n * 8);
    memcpy((uint8_t *)c->h_gmap_stage + idx_bytes, h_slots, (size_t)n * sizeof(GroupSlot));
    uint8_t *st = (uint8_t *)c->d_gmap_stage.p;
    HIPCHK(c, hipMemcpyAsync(st, c->h_gmap_stage, need, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(gmap_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, (GroupSlot *)c->d_gmap.p, (const uint64_t *)st,
                       (const GroupSlot *)(st + idx_bytes), n);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->gmap_ev, c->stream));
    c->gmap_ev_valid = true;
    return AGC_HIP_OK;
}

} // extern "C"

namespace {

// a piece of the work area
template <typename T> T *carve(uint8_t *&p, size_t n)
{
    T *r = (T *)p;
    p += (n * sizeof(T) + 255) & ~(size_t)255;
    return r;
}

} // namespace

// launches, on the second lane, the encode of the segments of the last agc_hip_segments_packed call whose group the table knew
static int launch_known(agc_hip_ctx *c)
{
    agc_hip_ctx::SegState &S = c->seg_state;
    const uint32_t n_ub = S.n_ub;
    const hipStream_t st = c->stream;
    DevSeg *segs = (DevSeg *)S.segs;
    SegCounts *counts = (SegCounts *)S.counts;
    uint32_t *flag = (uint32_t *)S.flag, *known_rank = (uint32_t *)S.known_rank;
    unsigned long long *capv = (unsigned long long *)S.capv, *cap_off = (unsigned long long *)S.cap_off;
    SegDesc *descs = (SegDesc *)S.descs;
    CHK(upload_refs(c));
    // a slot = len + 5 len / 16 + 64, rounded up to 16 (known_flag_kernel), and neighbouring segments share k <= 32 symbols:
    // per segment 32 * 21 / 16 + 64 + 15 < 128 bytes beyond its share of the sample
    const size_t scratch_ub = (size_t)(S.total + 5 * S.total / 16) + 128 * (size_t)n_ub + 64;
    CHK(ensure(c, c->l2.d_segs, (size_t)n_ub * sizeof(SegDesc), c->stream2));
    CHK(ensure(c, c->l2.d_resv, (size_t)n_ub * 4, c->stream2));
    CHK(ensure(c, c->l2.d_resp, (size_t)n_ub * 4, c->stream2));
    CHK(ensure(c, c->l2.d_scratch, scratch_ub, c->stream2));
    CHK(ensure(c, c->l2.d_n, 64, c->stream2));
    if (c->l2.h_lens_cap < n_ub) {
        if (c->l2.h_lens)
            HIPCHK(c, hipHostFree(c->l2.h_lens));
        c->l2.h_lens = nullptr;
        c->l2.h_lens_cap = 0;
        HIPCHK(c, hipHostMalloc((void **)&c->l2.h_lens, ((size_t)n_ub + n_ub / 4 + 1024) * 4, hipHostMallocDefault));
        c->l2.h_lens_cap = (size_t)n_ub + n_ub / 4 + 1024;
    }
    hipLaunchKernelGGL(known_flag_kernel, dim3((n_ub + 256) / 256), dim3(256), 0, st, segs, counts, (const RefDesc *)c->d_refs.p, (uint32_t)c->refs.size(), n_ub, flag,
                       capv);
    hipLaunchKernelGGL(scan_excl_kernel<uint32_t>, dim3(1), dim3(1024), 0, st, flag, known_rank, n_ub + 1);
    hipLaunchKernelGGL(scan_excl_kernel<unsigned long long>, dim3(1), dim3(1024), 0, st, capv, cap_off, n_ub + 1);
    const PackedView pv = {S.pk.d_words, S.pk.d_esc_index, S.pk.d_esc_bytes, S.pk.n_symbols};
    hipLaunchKernelGGL(known_emit_kernel, dim3((n_ub + 255) / 256), dim3(256), 0, st, segs, counts, flag, known_rank, cap_off, n_ub, pv, (const uint64_t *)S.d_ctg_off,
                       descs);
    // longest first (to the length bucket): count, scan, scatter
    uint32_t *lcnt = (uint32_t *)S.lcnt, *lstart = lcnt + (LEN_BUCKETS + 1), *lcur = lstart + (LEN_BUCKETS + 1);
    HIPCHK(c, hipMemsetAsync(lcnt, 0, (size_t)(3 * (LEN_BUCKETS + 1)) * 4, st));
    hipLaunchKernelGGL(known_len_count_kernel, dim3((n_ub + 255) / 256), dim3(256), 0, st, descs, counts, lcnt);
    hipLaunchKernelGGL(scan_excl_kernel<uint32_t>, dim3(1), dim3(1024), 0, st, lcnt, lstart, LEN_BUCKETS + 1);
    hipLaunchKernelGGL(known_order_kernel, dim3((n_ub + 255) / 256), dim3(256), 0, st, descs, counts, lstart, lcur, (SegDesc *)c->l2.d_segs.p);
    HIPCHK(c, hipGetLastError());
    // the number of descriptors moves into a word the lane owns: the next sample's agc_hip_segments_packed clears and re-lays-out
    // the work area on the first stream while this parse (and the copy of the count behind it) may still be running
    uint32_t *d_n_known = (uint32_t *)c->l2.d_n.p;
    HIPCHK(c, hipMemcpyAsync(d_n_known, &counts->n_known, 4, hipMemcpyDeviceToDevice, st));
    // the parse on the second lane, behind everything queued here
    HIPCHK(c, hipEventRecord(c->l2.ready, st));
    HIPCHK(c, hipStreamWaitEvent(c->stream2, c->l2.ready, 0));
    c->l2.timed = c->timing;
    if (c->l2.timed)
        (void)hipEventRecord(c->l2.e0, c->stream2);
    CHK(launch_parse<MODE_ENCODE>(c, n_ub, (uint8_t *)c->l2.d_scratch.p, nullptr, 1, d_n_known));
    if (c->l2.timed)
        (void)hipEventRecord(c->l2.e1, c->stream2);
    HIPCHK(c, hipEventRecord(c->l2.done, c->stream2));
    c->l2.done_valid = true;
    HIPCHK(c, hipMemcpyAsync(c->l2.h_lens, c->l2.d_resv.p, (size_t)n_ub * 4, hipMemcpyDeviceToHost, c->stream2));
    uint32_t *n_pinned = (uint32_t *)((uint8_t *)c->h_segcounts + 64);
    HIPCHK(c, hipMemcpyAsync(n_pinned, d_n_known, 4, hipMemcpyDeviceToHost, c->stream2));
    c->l2.n_pinned = n_pinned;
    c->l2.n = 0;
    c->l2.pending = true;
    return AGC_HIP_OK;
}

extern "C" {

int agc_hip_segments_packed(agc_hip_ctx *c, const agc_hip_packed *pk, const uint64_t *h_ctg_off, uint32_t n_ctg, uint32_t k, int prefetched,
                            int encode_known, uint64_t cap, agc_hip_segment *h_segs, uint64_t *h_n_segs, uint32_t *h_n_encoded)
{
    static_assert(sizeof(agc_hip_segment) == sizeof(DevSeg) && sizeof(DevSeg) == 56, "segment layout");
    if (!c || !pk || !h_ctg_off || !h_n_segs || k < 16 || k > 32 || (cap && !h_segs))
        return AGC_HIP_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    *h_n_segs = 0;
    if (h_n_encoded)
        *h_n_encoded = 0;
    c->seg_state.valid = false;
    if (!n_ctg)
        return AGC_HIP_OK;
    if (!pk->d_words || !pk->d_esc_index || h_ctg_off[n_ctg] > pk->n_symbols)
        return AGC_HIP_EINVAL;
    if (encode_known && c->l2.pending) {
        c->err = "segments_packed: the previous encode was not collected (agc_hip_lz_encode_end)";
        return AGC_HIP_EINVAL;
    }
    static const bool laps = getenv("AGC_HIP_LAPS") != nullptr;
    auto tnow = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double lt = laps ? tnow() : 0;
    auto LAP = [&](const char *what) {
        if (laps) {
            const double t = tnow();
            fprintf(stderr, "    segments_packed lap %s %.3f ms\n", what, t - lt);
            lt = t;
        }
    };
    // ---- raw hits: collected from the prefetch, or scanned now
    uint32_t n = 0;
    const ScanHit *d_hits = nullptr;
    agc_hip_ctx::Prefetch &pf = c->pf;
    bool from_pf = false;
    if (prefetched) {
        if (!pf.valid || pf.words != pk->d_words || pf.n_symbols != pk->n_symbols || pf.k != k || pf.n_ctg != n_ctg || pf.first_off != h_ctg_off[0] ||
            pf.last_off != h_ctg_off[n_ctg])
            return AGC_HIP_EINVAL; // nothing, or something else, was prefetched
        HIPCHK(c, hipStreamSynchronize(pf.stream));
        if (pf.timed) {
            float ms = 0;
            (void)hipEventElapsedTime(&ms, pf.e0, pf.e1);
            c->ms[AGC_HIP_K_SCAN] += ms;
            c->launches[AGC_HIP_K_SCAN] += 1;
            pf.timed = false;
        }
        if (pf.scanned && *pf.h_count <= pf.dev_cap) {
            n = *pf.h_count;
            d_hits = (const ScanHit *)pf.d_hits.p;
            from_pf = true;
        }
    }
    if (!from_pf) {
        if (!prefetched || pf.scanned) // (a prefetch whose hit list overflowed: the scan again, the ordinary way)
            CHK(packed_scan_raw(c, pk, h_ctg_off, n_ctg, k, &n));
        d_hits = (const ScanHit *)c->d_hits.p;
    }
    LAP("hits");
    const uint64_t n_ub64 = (uint64_t)n + n_ctg;
    if (n_ub64 > 0x3fffffffULL)
        return AGC_HIP_EINVAL;
    const uint32_t n_ub = (uint32_t)n_ub64;
    if (cap < n_ub) {
        *h_n_segs = n_ub;
        return AGC_HIP_ECAP; // (a prefetched scan stays valid: the second call collects it again)
    }
    if (!c->h_segcounts)
        HIPCHK(c, hipHostMalloc(&c->h_segcounts, 256, hipHostMallocDefault));
    // ---- work area
    auto layout = [&](uint8_t *p, bool assign) -> size_t {
        uint8_t *p0 = p;
        agc_hip_ctx::SegState &S = c->seg_state;
        auto take_ = [&](void **dst, size_t bytes) {
            if (assign && dst)
                *dst = p;
            p += (bytes + 255) & ~(size_t)255;
        };
        take_(&S.segs, (size_t)n_ub * sizeof(DevSeg));
        take_(&S.counts, sizeof(SegCounts));
        take_(&S.d_ctg_off, ((size_t)n_ctg + 1) * 8);
        take_(&S.flag, ((size_t)n_ub + 1) * 4);
        take_(&S.capv, ((size_t)n_ub + 1) * 8);
        take_(&S.known_rank, ((size_t)n_ub + 1) * 4);
        take_(&S.cap_off, ((size_t)n_ub + 1) * 8);
        take_(&S.descs, (size_t)n_ub * sizeof(SegDesc));
        take_(&S.lcnt, (size_t)(3 * (LEN_BUCKETS + 1)) * 4);
        return (size_t)(p - p0);
    };
    const size_t fixed_bytes = layout(nullptr, false);
    size_t hit_bytes = 0;
    {
        uint8_t *p = nullptr;
        carve<uint64_t>(p, n), carve<uint32_t>(p, n), carve<uint32_t>(p, 3 * ((size_t)2 * n + 1030));                 // sorted hits, the sort's buckets
        carve<uint32_t>(p, n), carve<uint32_t>(p, (size_t)n + 1), carve<uint32_t>(p, (size_t)n + 1), carve<uint32_t>(p, n); // ctg, take, acc_rank, acc_idx
        carve<uint32_t>(p, (size_t)n_ctg + 1), carve<uint32_t>(p, (size_t)n_ctg + 1);
        hit_bytes = (size_t)(p - (uint8_t *)nullptr);
    }
    CHK(ensure(c, c->d_segwork, fixed_bytes + hit_bytes + 256));
    layout((uint8_t *)c->d_segwork.p, true);
    agc_hip_ctx::SegState &S = c->seg_state;
    uint8_t *p = (uint8_t *)c->d_segwork.p + fixed_bytes;
    uint64_t *keys0 = carve<uint64_t>(p, n);
    uint32_t *vals0 = carve<uint32_t>(p, n), *bwork = carve<uint32_t>(p, 3 * ((size_t)2 * n + 1030));
    uint32_t *ctg = carve<uint32_t>(p, n), *take = carve<uint32_t>(p, (size_t)n + 1), *acc_rank = carve<uint32_t>(p, (size_t)n + 1), *acc_idx = carve<uint32_t>(p, n);
    uint32_t *hits_before = carve<uint32_t>(p, (size_t)n_ctg + 1), *tails_before = carve<uint32_t>(p, (size_t)n_ctg + 1);
    DevSeg *segs = (DevSeg *)S.segs;
    SegCounts *counts = (SegCounts *)S.counts;
    uint64_t *d_ctg_off = (uint64_t *)S.d_ctg_off;
    const hipStream_t st = c->stream;

    CHK(upload(c, d_ctg_off, h_ctg_off, ((size_t)n_ctg + 1) * 8, st));
    HIPCHK(c, hipMemsetAsync(counts, 0, sizeof(SegCounts), st));
    const uint64_t *pos_sorted = keys0;
    const uint32_t *order = vals0;
    KTimer tm(c, AGC_HIP_K_SEGMENTS);
    if (n) {
        // position order: counting sort into ~n / 2 position buckets, a handful of hits each (seg_kernels.hip)
        const uint64_t base = h_ctg_off[0], span = h_ctg_off[n_ctg] - base;
        uint64_t nb_target = 1024;
        while (nb_target < n / 2)
            nb_target <<= 1;
        uint32_t sh = 0;
        while ((span >> sh) >= nb_target)
            ++sh;
        const uint32_t n_buckets = (uint32_t)(span >> sh) + 1; // <= nb_target <= max(1024, n)
        uint32_t *bcnt = bwork, *bstart = bwork + ((size_t)2 * n + 1030), *bcur = bstart + ((size_t)2 * n + 1030);
        HIPCHK(c, hipMemsetAsync(bcnt, 0, ((size_t)n_buckets + 1) * 4, st));
        HIPCHK(c, hipMemsetAsync(bcur, 0, ((size_t)n_buckets + 1) * 4, st));
        hipLaunchKernelGGL(hb_count_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_hits, n, base, sh, bcnt);
        hipLaunchKernelGGL(scan_excl_kernel<uint32_t>, dim3(1), dim3(1024), 0, st, bcnt, bstart, n_buckets + 1);
        hipLaunchKernelGGL(hb_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_hits, n, base, sh, bstart, bcur, keys0, vals0);
        hipLaunchKernelGGL(hb_sort_kernel, dim3((n_buckets + 255) / 256), dim3(256), 0, st, bstart, n_buckets, keys0, vals0);
        hipLaunchKernelGGL(hit_contig_kernel, dim3((n + 255) / 256), dim3(256), 0, st, pos_sorted, n, d_ctg_off, n_ctg, ctg);
        hipLaunchKernelGGL(hit_accept_kernel, dim3((n + 256) / 256), dim3(256), 0, st, pos_sorted, ctg, n, k, take);
        hipLaunchKernelGGL(scan_excl_kernel<uint32_t>, dim3(1), dim3(1024), 0, st, take, acc_rank, n + 1);
        hipLaunchKernelGGL(hit_compact_kernel, dim3((n + 255) / 256), dim3(256), 0, st, take, acc_rank, n, acc_idx);
    }
    hipLaunchKernelGGL(contig_sums_kernel, dim3(1), dim3(1024), 0, st, d_ctg_off, n_ctg, k, pos_sorted, ctg, acc_idx, acc_rank, n, hits_before, tails_before, counts);
    hipLaunchKernelGGL(seg_tail_kernel, dim3((n_ctg + 255) / 256), dim3(256), 0, st, d_ctg_off, n_ctg, hits_before, tails_before, segs);
    if (n)
        hipLaunchKernelGGL(seg_fill_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_hits, order, ctg, acc_idx, counts, k, d_ctg_off, hits_before, tails_before, segs);
    HIPCHK(c, hipGetLastError());
    // ---- the look-up (LDS-staged bucket ranges when the batch is large against the table, straight from HBM otherwise)
    if (c->gmap_slots) {
        const uint64_t mask = c->gmap_slots - 1;
        const uint64_t range_slots = std::min<uint64_t>(c->gmap_slots, GM_RANGE_SLOTS);
        const uint64_t n_ranges = c->gmap_slots / range_slots;
        const bool staged = n_ranges <= 65535 && (uint64_t)n_ub * 8 >= c->gmap_slots;
        if (staged) {
            if (!c->lds_lookup_set) {
                HIPCHK(c, hipFuncSetAttribute((const void *)group_lookup_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GM_RANGE_SLOTS * sizeof(GroupSlot)));
                c->lds_lookup_set = true;
            }
            hipLaunchKernelGGL(group_lookup_kernel, dim3((uint32_t)n_ranges), dim3(1024), range_slots * sizeof(GroupSlot), st, (const GroupSlot *)c->d_gmap.p, mask, segs,
                               counts, 1u);
        } else
            hipLaunchKernelGGL(group_lookup_kernel, dim3(std::max(1u, std::min(1024u, (n_ub + 1023) / 1024))), dim3(1024), 0, st, (const GroupSlot *)c->d_gmap.p, mask,
                               segs, counts, 0u);
        HIPCHK(c, hipGetLastError());
    }
    LAP("cut + look-up queued");
    S.n_ub = n_ub;
    S.total = h_ctg_off[n_ctg] - h_ctg_off[0];
    S.pk = *pk;
    S.valid = c->gmap_slots != 0 && n_ub != 0;
    // ---- the encode of the segments whose group is known, launched from here (second lane) when asked
    bool launched = false;
    if (encode_known && S.valid) {
        CHK(launch_known(c));
        launched = true;
        LAP("encode queued");
    }
    // ---- the segment table to the host (into pinned memory when the caller's buffer is: agc_hip_host_alloc)
    HIPCHK(c, hipMemcpyAsync(c->h_segcounts, counts, sizeof(SegCounts), hipMemcpyDeviceToHost, st));
    if (n_ub)
        HIPCHK(c, hipMemcpyAsync(h_segs, segs, (size_t)n_ub * sizeof(DevSeg), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    LAP("segments on the host");
    const SegCounts hc = *(const SegCounts *)c->h_segcounts;
    *h_n_segs = hc.n_segs;
    if (launched) {
        // (the flags `encoded` were set before the copy: their number is the number of deltas; the device's own count arrives
        // behind the parse and is what agc_hip_lz_encode_end goes by)
        uint32_t ne = 0;
        for (uint32_t i = 0; i < hc.n_segs; ++i)
            ne += h_segs[i].encoded != 0;
        if (h_n_encoded)
            *h_n_encoded = ne;
        if (!ne) { // (no group was known: the launch had nothing to do and nothing is in flight)
            HIPCHK(c, hipStreamSynchronize(c->stream2));
            c->l2.n_pinned = nullptr;
            c->l2.pending = false;
        }
    }
    if (from_pf)
        pf.valid = false;
    return AGC_HIP_OK;
}

// The same launch as a call of its own, for a caller that wants the segment table first (include/agc_hip.h)
int agc_hip_segments_encode_known(agc_hip_ctx *c)
{
    if (!c)
        return AGC_HIP_EINVAL;
    if (!c->seg_state.valid) {
        c->err = "segments_encode_known: no segments on the device (agc_hip_segments_packed comes first)";
        return AGC_HIP_EINVAL;
    }
    if (c->l2.pending) {
        c->err = "segments_encode_known: the previous encode was not collected (agc_hip_lz_encode_end)";
        return AGC_HIP_EINVAL;
    }
    HIPCHK(c, hipSetDevice(c->device));
    c->seg_state.valid = false;
    return launch_known(c);
}

} // extern "C"
