// async_encode.hip -- asynchronous form of agc_hip_lz_encode_batch_dev (included at the end of api.hip).
//
// begin: descriptors, reverse-complement staging and the parse kernel go to a SECOND stream with their own buffer set and
// the call returns; end: waits, compacts the deltas and copies them to the host.  Between the two the context's other
// entry points (estimates, split points, scans: first stream, first buffer set) run concurrently -- the low-occupancy
// cost-vector kernel (~3000 waves, one wave's latency long) then hides behind the encode of the segments whose group is
// already known.  No reference may be registered between begin and end (the descriptor table must not move).
// Same kernels, same results as the synchronous call.
namespace {

struct AsyncEnc {
    hipStream_t stream = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    DevBuf d_stage, d_slices, d_segs, d_scratch, d_resv, d_resp, d_dstoff, d_compact;
    std::vector<SliceDesc> h_slices;
    std::vector<SegDesc> h_segs;
    uint32_t n = 0;
    bool in_flight = false;
};

int ensure2(agc_hip_ctx *c, AsyncEnc *a, DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap)
        return AGC_HIP_OK;
    size_t want = std::max(bytes + bytes / 4, b.cap + b.cap / 2);
    want = (want + 255) & ~(size_t)255;
    if (b.p) {
        HIPCHK(c, hipStreamSynchronize(a->stream));
        HIPCHK(c, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    HIPCHK(c, hipMalloc(&b.p, want));
    b.cap = want;
    return AGC_HIP_OK;
}

} // namespace

extern "C" {

int agc_hip_lz_encode_begin_dev(agc_hip_ctx *c, uint32_t n, const uint32_t *h_gid, const uint8_t *d_base, const uint64_t *h_off,
                                const uint32_t *h_len, const uint8_t *h_rc)
{
    if (!c || (n && (!h_gid || !h_off || !h_len || !d_base)))
        return AGC_HIP_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->async_enc) {
        AsyncEnc *a = new AsyncEnc();
        if (hipStreamCreateWithFlags(&a->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&a->e0) != hipSuccess ||
            hipEventCreate(&a->e1) != hipSuccess) {
            delete a;
            c->err = "cannot create the second stream";
            return AGC_HIP_ENODEV;
        }
        c->async_enc = a;
    }
    AsyncEnc *a = (AsyncEnc *)c->async_enc;
    if (a->in_flight) {
        c->err = "an asynchronous encode is already in flight";
        return AGC_HIP_EINVAL;
    }
    a->n = n;
    a->in_flight = true;
    if (!n)
        return AGC_HIP_OK;
    for (uint32_t i = 0; i < n; ++i)
        if (h_gid[i] >= c->refs.size() || !c->refs[h_gid[i]].valid) {
            a->in_flight = false;
            c->err = "group " + std::to_string(h_gid[i]) + " has no registered reference";
            return AGC_HIP_ENOREF;
        }
    CHK(upload_refs(c)); // (synchronises the first stream: the table is in place before the second stream reads it)
    // staging of the reverse-complemented texts
    size_t stage = 0;
    std::vector<size_t> soff(n, 0);
    uint32_t n_rc = 0;
    for (uint32_t i = 0; i < n; ++i)
        if (h_rc && h_rc[i]) {
            soff[i] = stage;
            stage += ((size_t)h_len[i] + 15) & ~(size_t)15;
            ++n_rc;
        }
    if (n_rc) {
        CHK(ensure2(c, a, a->d_stage, stage + 64));
        a->h_slices.clear();
        a->h_slices.reserve(n_rc);
        for (uint32_t i = 0; i < n; ++i)
            if (h_rc[i])
                a->h_slices.push_back({d_base + h_off[i], (uint8_t *)a->d_stage.p + soff[i], h_len[i], 1u, 0u, 0u});
        CHK(ensure2(c, a, a->d_slices, a->h_slices.size() * sizeof(SliceDesc)));
        HIPCHK(c, hipMemcpyAsync(a->d_slices.p, a->h_slices.data(), a->h_slices.size() * sizeof(SliceDesc), hipMemcpyHostToDevice, a->stream));
        hipLaunchKernelGGL(slice_copy_kernel, dim3(grid_for(n_rc, 1, 65536)), dim3(256), 0, a->stream, (const SliceDesc *)a->d_slices.p, n_rc);
        HIPCHK(c, hipGetLastError());
    }
    // longest first, as the synchronous call
    std::vector<uint64_t> keys(n);
    for (uint32_t i = 0; i < n; ++i)
        keys[i] = ((uint64_t)(~h_len[i]) << 32) | i;
    std::sort(keys.begin(), keys.end());
    std::vector<uint64_t> ooff(n);
    uint64_t tot = 0;
    for (uint32_t i = 0; i < n; ++i) {
        ooff[i] = tot;
        tot += (((uint64_t)h_len[i] + 5ULL * h_len[i] / 16 + 64) + 15) & ~15ULL;
    }
    a->h_segs.resize(n);
    for (uint32_t p = 0; p < n; ++p) {
        const uint32_t i = (uint32_t)keys[p];
        SegDesc &s = a->h_segs[p];
        s.text = (h_rc && h_rc[i]) ? (const uint8_t *)a->d_stage.p + soff[i] : d_base + h_off[i];
        s.out_off = ooff[i];
        s.len = h_len[i];
        s.ref_slot = h_gid[i];
        s.flags = 0;
        s.pad = i;
    }
    CHK(ensure2(c, a, a->d_segs, (size_t)n * sizeof(SegDesc)));
    CHK(ensure2(c, a, a->d_resv, (size_t)n * 4));
    CHK(ensure2(c, a, a->d_resp, (size_t)n * 4));
    CHK(ensure2(c, a, a->d_scratch, tot + 64));
    HIPCHK(c, hipMemcpyAsync(a->d_segs.p, a->h_segs.data(), (size_t)n * sizeof(SegDesc), hipMemcpyHostToDevice, a->stream));
    if (c->timing)
        (void)hipEventRecord(a->e0, a->stream);
    hipLaunchKernelGGL(lz_parse_kernel<MODE_ENCODE>, dim3((n + 3) / 4), dim3(256), 0, a->stream, (const RefDesc *)c->d_refs.p,
                       (const SegDesc *)a->d_segs.p, n, (uint8_t *)a->d_scratch.p, (uint32_t *)nullptr, (uint32_t *)a->d_resv.p,
                       (uint32_t *)a->d_resp.p);
    HIPCHK(c, hipGetLastError());
    if (c->timing)
        (void)hipEventRecord(a->e1, a->stream);
    return AGC_HIP_OK;
}

int agc_hip_lz_encode_end(agc_hip_ctx *c, uint8_t *h_enc, uint64_t enc_cap, uint64_t *h_enc_off)
{
    if (!c || !h_enc_off || !c->async_enc)
        return AGC_HIP_EINVAL;
    AsyncEnc *a = (AsyncEnc *)c->async_enc;
    if (!a->in_flight) {
        c->err = "no asynchronous encode in flight";
        return AGC_HIP_EINVAL;
    }
    const uint32_t n = a->n;
    h_enc_off[0] = 0;
    if (!n) {
        a->in_flight = false;
        return AGC_HIP_OK;
    }
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<uint32_t> lens(n);
    HIPCHK(c, hipMemcpyAsync(lens.data(), a->d_resv.p, (size_t)n * 4, hipMemcpyDeviceToHost, a->stream));
    HIPCHK(c, hipStreamSynchronize(a->stream));
    if (c->timing) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, a->e0, a->e1) == hipSuccess) {
            c->ms[AGC_HIP_K_ENCODE] += ms;
            c->launches[AGC_HIP_K_ENCODE] += 1;
        }
    }
    for (uint32_t i = 0; i < n; ++i)
        h_enc_off[i + 1] = h_enc_off[i] + lens[i];
    const uint64_t tot = h_enc_off[n];
    if (tot > enc_cap)
        return AGC_HIP_ECAP; // (still in flight: call again with a buffer of h_enc_off[n] bytes)
    a->in_flight = false;
    if (!tot)
        return AGC_HIP_OK;
    if (!h_enc)
        return AGC_HIP_EINVAL;
    CHK(ensure2(c, a, a->d_dstoff, (size_t)n * 8));
    CHK(ensure2(c, a, a->d_compact, tot));
    HIPCHK(c, hipMemcpyAsync(a->d_dstoff.p, h_enc_off, (size_t)n * 8, hipMemcpyHostToDevice, a->stream));
    hipLaunchKernelGGL(gather_bytes_kernel, dim3(grid_for(n, 1, 8192)), dim3(256), 0, a->stream, (const uint8_t *)a->d_scratch.p,
                       (const SegDesc *)a->d_segs.p, (const uint32_t *)a->d_resv.p, (const uint64_t *)a->d_dstoff.p, n,
                       (uint8_t *)a->d_compact.p);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(h_enc, a->d_compact.p, tot, hipMemcpyDeviceToHost, a->stream));
    HIPCHK(c, hipStreamSynchronize(a->stream));
    return AGC_HIP_OK;
}

} // extern "C"

namespace {
void async_enc_destroy(agc_hip_ctx *c)
{
    AsyncEnc *a = (AsyncEnc *)c->async_enc;
    if (!a)
        return;
    (void)hipStreamSynchronize(a->stream);
    DevBuf *bufs[] = {&a->d_stage, &a->d_slices, &a->d_segs, &a->d_scratch, &a->d_resv, &a->d_resp, &a->d_dstoff, &a->d_compact};
    for (DevBuf *b : bufs)
        if (b->p)
            (void)hipFree(b->p);
    (void)hipEventDestroy(a->e0);
    (void)hipEventDestroy(a->e1);
    (void)hipStreamDestroy(a->stream);
    delete a;
    c->async_enc = nullptr;
}
} // namespace
