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
    const size_t idx_bytes = ((size_t)n * 8 + 15) & ~(size_t)15;
    CHK(ensure(c, c->d_gmap_stage, idx_bytes + (size_t)n * sizeof(GroupSlot) + 64));
    uint8_t *st = (uint8_t *)c->d_gmap_stage.p;
    HIPCHK(c, hipMemcpyAsync(st, h_idx, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(st + idx_bytes, h_slots, (size_t)n * sizeof(GroupSlot), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(gmap_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, (GroupSlot *)c->d_gmap.p, (const uint64_t *)st,
                       (const GroupSlot *)(st + idx_bytes), n);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
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
    if (!n_ctg)
        return AGC_HIP_OK;
    if (!pk->d_words || !pk->d_esc_index || h_ctg_off[n_ctg] > pk->n_symbols)
        return AGC_HIP_EINVAL;
    if (encode_known && c->l2.pending) {
        c->err = "segments_packed: the previous encode was not collected (agc_hip_lz_encode_end)";
        return AGC_HIP_EINVAL;
    }
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
    size_t need = 0;
    {
        uint8_t *p = nullptr;
        carve<uint64_t>(p, n), carve<uint64_t>(p, n), carve<uint32_t>(p, n), carve<uint32_t>(p, n);           // sort buffers of the hits
        carve<uint32_t>(p, n), carve<uint32_t>(p, n), carve<uint32_t>(p, (size_t)n + 1), carve<uint32_t>(p, n); // ctg, take, acc_rank, acc_idx
        carve<uint64_t>(p, (size_t)n_ctg + 1), carve<uint32_t>(p, n_ctg), carve<unsigned long long>(p, n_ctg);
        carve<uint32_t>(p, (size_t)n_ctg + 1), carve<uint32_t>(p, (size_t)n_ctg + 1);
        carve<DevSeg>(p, n_ub), carve<SegCounts>(p, 1);
        carve<uint32_t>(p, (size_t)n_ub + 1), carve<unsigned long long>(p, (size_t)n_ub + 1), carve<uint32_t>(p, (size_t)n_ub + 1),
            carve<unsigned long long>(p, (size_t)n_ub + 1);
        carve<SegDesc>(p, n_ub), carve<uint32_t>(p, n_ub), carve<uint32_t>(p, n_ub), carve<uint32_t>(p, n_ub), carve<uint32_t>(p, n_ub);
        need = (size_t)(p - (uint8_t *)nullptr);
    }
    CHK(ensure(c, c->d_segwork, need + 256));
    uint8_t *p = (uint8_t *)c->d_segwork.p;
    uint64_t *keys0 = carve<uint64_t>(p, n), *keys1 = carve<uint64_t>(p, n);
    uint32_t *vals0 = carve<uint32_t>(p, n), *vals1 = carve<uint32_t>(p, n);
    uint32_t *ctg = carve<uint32_t>(p, n), *take = carve<uint32_t>(p, n), *acc_rank = carve<uint32_t>(p, (size_t)n + 1), *acc_idx = carve<uint32_t>(p, n);
    uint64_t *d_ctg_off = carve<uint64_t>(p, (size_t)n_ctg + 1);
    uint32_t *per_ctg = carve<uint32_t>(p, n_ctg);
    unsigned long long *last_pos = carve<unsigned long long>(p, n_ctg);
    uint32_t *hits_before = carve<uint32_t>(p, (size_t)n_ctg + 1), *tails_before = carve<uint32_t>(p, (size_t)n_ctg + 1);
    DevSeg *segs = carve<DevSeg>(p, n_ub);
    SegCounts *counts = carve<SegCounts>(p, 1);
    uint32_t *flag = carve<uint32_t>(p, (size_t)n_ub + 1);
    unsigned long long *capv = carve<unsigned long long>(p, (size_t)n_ub + 1);
    uint32_t *known_rank = carve<uint32_t>(p, (size_t)n_ub + 1);
    unsigned long long *cap_off = carve<unsigned long long>(p, (size_t)n_ub + 1);
    SegDesc *descs = carve<SegDesc>(p, n_ub);
    uint32_t *skey0 = carve<uint32_t>(p, n_ub), *skey1 = carve<uint32_t>(p, n_ub), *sval0 = carve<uint32_t>(p, n_ub), *sval1 = carve<uint32_t>(p, n_ub);
    const hipStream_t st = c->stream;
    auto tmp_for = [&](size_t bytes) -> int { return ensure(c, c->d_segtmp, bytes + 256); };

    HIPCHK(c, hipMemcpyAsync(d_ctg_off, h_ctg_off, ((size_t)n_ctg + 1) * 8, hipMemcpyHostToDevice, st));
    HIPCHK(c, hipMemsetAsync(per_ctg, 0, (size_t)n_ctg * 4, st));
    HIPCHK(c, hipMemsetAsync(last_pos, 0, (size_t)n_ctg * 8, st));
    HIPCHK(c, hipMemsetAsync(counts, 0, sizeof(SegCounts), st));
    const uint64_t *pos_sorted = keys0;
    const uint32_t *order = vals0;
    KTimer tm(c, AGC_HIP_K_SEGMENTS);
    if (n) {
        hipLaunchKernelGGL(hit_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_hits, n, keys0, vals0);
        rocprim::double_buffer<uint64_t> dk(keys0, keys1);
        rocprim::double_buffer<uint32_t> dv(vals0, vals1);
        size_t tb = 0;
        // (positions lie below 2^40: five 8-bit passes)
        if (rocprim::radix_sort_pairs(nullptr, tb, dk, dv, (size_t)n, 0, 40, st) != hipSuccess)
            return AGC_HIP_ENODEV;
        CHK(tmp_for(tb));
        if (rocprim::radix_sort_pairs(c->d_segtmp.p, tb, dk, dv, (size_t)n, 0, 40, st) != hipSuccess)
            return AGC_HIP_ENODEV;
        pos_sorted = dk.current();
        order = dv.current();
        hipLaunchKernelGGL(hit_contig_kernel, dim3((n + 255) / 256), dim3(256), 0, st, pos_sorted, n, d_ctg_off, n_ctg, ctg);
        hipLaunchKernelGGL(hit_accept_kernel, dim3((n + 255) / 256), dim3(256), 0, st, pos_sorted, ctg, n, k, take, per_ctg, last_pos);
        tb = 0;
        if (rocprim::exclusive_scan(nullptr, tb, take, acc_rank, 0u, (size_t)n, rocprim::plus<uint32_t>(), st) != hipSuccess)
            return AGC_HIP_ENODEV;
        CHK(tmp_for(tb));
        if (rocprim::exclusive_scan(c->d_segtmp.p, tb, take, acc_rank, 0u, (size_t)n, rocprim::plus<uint32_t>(), st) != hipSuccess)
            return AGC_HIP_ENODEV;
    }
    hipLaunchKernelGGL(contig_sums_kernel, dim3(1), dim3(1024), 0, st, d_ctg_off, n_ctg, k, per_ctg, last_pos, hits_before, tails_before, counts);
    {
        const uint32_t m = std::max(n, n_ctg);
        hipLaunchKernelGGL(seg_cut_kernel, dim3((m + 255) / 256), dim3(256), 0, st, take, acc_rank, n, d_ctg_off, n_ctg, hits_before, tails_before, acc_idx, segs);
    }
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
            static bool lds_set = false;
            if (!lds_set) {
                HIPCHK(c, hipFuncSetAttribute((const void *)group_lookup_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GM_RANGE_SLOTS * sizeof(GroupSlot)));
                lds_set = true;
            }
            hipLaunchKernelGGL(group_lookup_kernel, dim3((uint32_t)n_ranges), dim3(1024), range_slots * sizeof(GroupSlot), st, (const GroupSlot *)c->d_gmap.p, mask, segs,
                               counts, 1u);
        } else
            hipLaunchKernelGGL(group_lookup_kernel, dim3(std::max(1u, std::min(1024u, (n_ub + 1023) / 1024))), dim3(1024), 0, st, (const GroupSlot *)c->d_gmap.p, mask,
                               segs, counts, 0u);
        HIPCHK(c, hipGetLastError());
    }
    // ---- the encode of the segments whose group is known, launched from here (second lane)
    const uint64_t total = h_ctg_off[n_ctg] - h_ctg_off[0];
    bool launched = false;
    if (encode_known && c->gmap_slots && n_ub) {
        CHK(upload_refs(c));
        const size_t scratch_ub = (size_t)(total + 5 * total / 16) + 96 * (size_t)n_ub + 64;
        CHK(ensure(c, c->l2.d_segs, (size_t)n_ub * sizeof(SegDesc), c->stream2));
        CHK(ensure(c, c->l2.d_resv, (size_t)n_ub * 4, c->stream2));
        CHK(ensure(c, c->l2.d_resp, (size_t)n_ub * 4, c->stream2));
        CHK(ensure(c, c->l2.d_scratch, scratch_ub, c->stream2));
        if (c->l2.h_lens_cap < n_ub) {
            if (c->l2.h_lens)
                HIPCHK(c, hipHostFree(c->l2.h_lens));
            c->l2.h_lens = nullptr;
            c->l2.h_lens_cap = 0;
            HIPCHK(c, hipHostMalloc((void **)&c->l2.h_lens, ((size_t)n_ub + n_ub / 4 + 1024) * 4, hipHostMallocDefault));
            c->l2.h_lens_cap = (size_t)n_ub + n_ub / 4 + 1024;
        }
        hipLaunchKernelGGL(known_flag_kernel, dim3((n_ub + 256) / 256), dim3(256), 0, st, segs, counts, (const RefDesc *)c->d_refs.p, (uint32_t)c->refs.size(), n_ub,
                           flag, capv);
        size_t tb = 0, tb2 = 0;
        if (rocprim::exclusive_scan(nullptr, tb, flag, known_rank, 0u, (size_t)n_ub + 1, rocprim::plus<uint32_t>(), st) != hipSuccess ||
            rocprim::exclusive_scan(nullptr, tb2, capv, cap_off, 0ull, (size_t)n_ub + 1, rocprim::plus<unsigned long long>(), st) != hipSuccess)
            return AGC_HIP_ENODEV;
        CHK(tmp_for(std::max(tb, tb2)));
        if (rocprim::exclusive_scan(c->d_segtmp.p, tb, flag, known_rank, 0u, (size_t)n_ub + 1, rocprim::plus<uint32_t>(), st) != hipSuccess ||
            rocprim::exclusive_scan(c->d_segtmp.p, tb2, capv, cap_off, 0ull, (size_t)n_ub + 1, rocprim::plus<unsigned long long>(), st) != hipSuccess)
            return AGC_HIP_ENODEV;
        HIPCHK(c, hipMemsetAsync(skey0, 0xFF, (size_t)n_ub * 4, st));
        const PackedView pv = {pk->d_words, pk->d_esc_index, pk->d_esc_bytes, pk->n_symbols};
        hipLaunchKernelGGL(known_emit_kernel, dim3((n_ub + 255) / 256), dim3(256), 0, st, segs, counts, flag, known_rank, cap_off, n_ub, pv, d_ctg_off, descs, skey0, sval0);
        rocprim::double_buffer<uint32_t> sk(skey0, skey1), sv(sval0, sval1);
        tb = 0;
        if (rocprim::radix_sort_pairs(nullptr, tb, sk, sv, (size_t)n_ub, 0, 32, st) != hipSuccess)
            return AGC_HIP_ENODEV;
        CHK(tmp_for(tb));
        if (rocprim::radix_sort_pairs(c->d_segtmp.p, tb, sk, sv, (size_t)n_ub, 0, 32, st) != hipSuccess)
            return AGC_HIP_ENODEV;
        hipLaunchKernelGGL(known_order_kernel, dim3((n_ub + 255) / 256), dim3(256), 0, st, descs, sv.current(), counts, (SegDesc *)c->l2.d_segs.p);
        HIPCHK(c, hipGetLastError());
        // the parse on the second lane, behind everything queued here
        HIPCHK(c, hipEventRecord(c->l2.ready, st));
        HIPCHK(c, hipStreamWaitEvent(c->stream2, c->l2.ready, 0));
        c->l2.timed = c->timing;
        if (c->l2.timed)
            (void)hipEventRecord(c->l2.e0, c->stream2);
        CHK(launch_parse<MODE_ENCODE>(c, n_ub, (uint8_t *)c->l2.d_scratch.p, nullptr, true, &counts->n_known));
        if (c->l2.timed)
            (void)hipEventRecord(c->l2.e1, c->stream2);
        HIPCHK(c, hipEventRecord(c->l2.done, c->stream2));
        c->l2.done_valid = true;
        HIPCHK(c, hipMemcpyAsync(c->l2.h_lens, c->l2.d_resv.p, (size_t)n_ub * 4, hipMemcpyDeviceToHost, c->stream2));
        launched = true;
    }
    // ---- the segment table to the host
    HIPCHK(c, hipMemcpyAsync(c->h_segcounts, counts, sizeof(SegCounts), hipMemcpyDeviceToHost, st));
    if (n_ub)
        HIPCHK(c, hipMemcpyAsync(h_segs, segs, (size_t)n_ub * sizeof(DevSeg), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    const SegCounts hc = *(const SegCounts *)c->h_segcounts;
    *h_n_segs = hc.n_segs;
    if (launched) {
        c->l2.n = hc.n_known;
        c->l2.pending = true;
        if (h_n_encoded)
            *h_n_encoded = hc.n_known;
    }
    if (from_pf)
        pf.valid = false;
    return AGC_HIP_OK;
}

} // extern "C"
