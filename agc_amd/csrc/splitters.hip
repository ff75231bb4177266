// splitters.hip -- reference-genome preprocessing on the GPU: determine_splitters
// (src/core/agc_compressor.cpp:428-563 with enumerate/remove_non_singletons/find_splitters_in_contig,
// :630-704, :762-825).  Once per archive, not the per-sample hot path, but at human
// scale the host version sorts 3 G k-mers for minutes; here: enumerate canonical k-mers (same packed
// window as the scan kernel) -> rocPRIM radix sort of (k-mer, position) pairs -> singleton positions
// as a bitmap -> the sequential "first singleton after >= segment_size symbols" walk on the host over
// that bitmap (~ n / segment_size jumps).  Included by api.hip (uses its context and helpers).
#include <rocprim/rocprim.hpp>

namespace agc {

// canonical k-mer ending at every position of the ranges (left-aligned u64, ~0 where the window
// holds a non-ACGT symbol or starts before the contig), written to keys[pos]; vals[pos] = pos
__global__ void __launch_bounds__(256) kmer_enum_kernel(const uint8_t *__restrict__ codes, const ScanRange *__restrict__ ranges,
                                                        uint32_t n_ranges, uint32_t k, uint64_t *__restrict__ keys,
                                                        uint32_t *__restrict__ vals)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t waves_per_block = blockDim.x >> 6;
    const uint64_t kmask = k == 32 ? ~0ULL : ((1ULL << (2 * k)) - 1ULL);
    const uint32_t lshift = 64 - 2 * k;
    const uint64_t wmask = k == 32 ? 0xFFFFFFFFULL : ((1ULL << k) - 1ULL);
    for (uint32_t r = blockIdx.x * waves_per_block + wave; r < n_ranges; r += gridDim.x * waves_per_block) {
        const ScanRange rg = ranges[r];
        for (uint64_t base = rg.begin; base < rg.end; base += 1024) {
            const uint64_t off = base + (uint64_t)lane * 16;
            if (off >= rg.end)
                continue;
            // symbols off-32 .. off+15 (invalid outside the contig); byte loads keep this rarely-run kernel simple
            uint64_t hi = 0, inv = 0;
            uint32_t P = 0;
#pragma unroll 1
            for (int q = 0; q < 48; ++q) {
                const int64_t p = (int64_t)off - 32 + q;
                uint32_t c = 4;
                if (p >= (int64_t)rg.ctg_begin && p < (int64_t)rg.ctg_end)
                    c = codes[p];
                if (q < 32)
                    hi = (hi << 2) | (c & 3);
                else
                    P = (P << 2) | (c & 3);
                inv = (inv << 1) | (c > 3);
            }
            const uint32_t nvalid = rg.end - off < 16 ? (uint32_t)(rg.end - off) : 16u;
            const uint64_t w_lo = (hi << 32) | P, w_hi = hi >> 32;
            for (uint32_t j = 0; j < nvalid; ++j) {
                uint64_t key = ~0ULL;
                if (((inv >> (15 - j)) & wmask) == 0) {
                    const uint32_t sft = 2 * (15 - j);
                    uint64_t dir = w_lo >> sft;
                    if (sft)
                        dir |= w_hi << (64 - sft);
                    dir &= kmask;
                    const uint64_t rcv = (rev2(~dir) >> lshift) & kmask;
                    const uint64_t dl = dir << lshift, rl = rcv << lshift;
                    key = dl < rl ? dl : rl;
                }
                keys[off + j] = key;
                vals[off + j] = (uint32_t)(off + j);
            }
        }
    }
}

// sorted (key, pos): a key that differs from both neighbours is a singleton -> set bit `pos`
__global__ void __launch_bounds__(256) singleton_bitmap_kernel(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                               uint64_t n, uint32_t *__restrict__ bitmap)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t x = keys[i];
        if (x == ~0ULL)
            continue;
        if ((i == 0 || keys[i - 1] != x) && (i + 1 == n || keys[i + 1] != x)) {
            const uint32_t p = vals[i];
            atomicOr(&bitmap[p >> 5], 1u << (p & 31));
        }
    }
}

// canonical k-mers ending at the given absolute positions (all known to be valid)
__global__ void __launch_bounds__(256) kmers_at_kernel(const uint8_t *__restrict__ codes, const uint64_t *__restrict__ pos, uint32_t n,
                                                       uint32_t k, uint64_t *__restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const uint64_t e = pos[i];
    uint64_t dir = 0, rc = 0;
    for (uint32_t t = 0; t < k; ++t) {
        const uint64_t c = codes[e - (k - 1) + t] & 3;
        dir = (dir << 2) | c;
        rc |= (3 - c) << (2 * t);
    }
    const uint32_t lshift = 64 - 2 * k;
    const uint64_t dl = dir << lshift, rl = rc << lshift;
    out[i] = dl < rl ? dl : rl;
}

} // namespace agc

extern "C" int agc_hip_determine_splitters_dev(agc_hip_ctx *c, const uint8_t *d_codes, const uint64_t *h_ctg_off, uint32_t n_ctg,
                                               uint32_t k, uint32_t segment_size, uint64_t cap, uint64_t *h_splitters,
                                               uint64_t *h_n_splitters, uint64_t sets_cap, uint64_t *h_sorted_kmers,
                                               uint64_t *h_n_sorted)
{
    using namespace agc;
    if (!c || !h_ctg_off || !h_n_splitters || k < 2 || k > 32 || segment_size == 0)
        return AGC_HIP_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    *h_n_splitters = 0;
    if (h_n_sorted)
        *h_n_sorted = 0;
    const uint64_t lo = n_ctg ? h_ctg_off[0] : 0, hi = n_ctg ? h_ctg_off[n_ctg] : 0;
    const uint64_t total = hi - lo;
    if (!total)
        return AGC_HIP_OK;
    if (!d_codes || hi > 0xFFFFFFFFULL)
        return AGC_HIP_EINVAL; // positions are carried as u32 through the sort

    std::vector<ScanRange> ranges;
    for (uint32_t ci = 0; ci < n_ctg; ++ci) {
        const uint64_t b = h_ctg_off[ci], e = h_ctg_off[ci + 1];
        if (e - b < k)
            continue;
        for (uint64_t p = b; p < e; p += 65536)
            ranges.push_back({b, e, p, std::min(e, p + 65536)});
    }
    if (ranges.empty())
        return AGC_HIP_OK;

    // buffers: keys/vals double-buffered for the sort, bitmap over absolute positions
    DevBuf keys0, keys1, vals0, vals1, tmp, bitmap;
    auto release = [&]() {
        for (DevBuf *b : {&keys0, &keys1, &vals0, &vals1, &tmp, &bitmap})
            if (b->p)
                (void)hipFree(b->p);
    };
    int rc = AGC_HIP_OK;
    auto fail = [&](int code) {
        release();
        return code;
    };
    if ((rc = ensure(c, keys0, hi * 8)) || (rc = ensure(c, keys1, hi * 8)) || (rc = ensure(c, vals0, hi * 4)) ||
        (rc = ensure(c, vals1, hi * 4)) || (rc = ensure(c, bitmap, (hi / 32 + 2) * 4)))
        return fail(rc);
    if (hipMemsetAsync(keys0.p, 0xFF, hi * 8, c->stream) != hipSuccess || hipMemsetAsync(vals0.p, 0, hi * 4, c->stream) != hipSuccess ||
        hipMemsetAsync(bitmap.p, 0, (hi / 32 + 2) * 4, c->stream) != hipSuccess)
        return fail(AGC_HIP_ENODEV);
    if ((rc = ensure(c, c->d_ranges, ranges.size() * sizeof(ScanRange))))
        return fail(rc);
    if (hipMemcpyAsync(c->d_ranges.p, ranges.data(), ranges.size() * sizeof(ScanRange), hipMemcpyHostToDevice, c->stream) != hipSuccess)
        return fail(AGC_HIP_ENODEV);
    {
        KTimer t(c, AGC_HIP_K_SCAN);
        hipLaunchKernelGGL(kmer_enum_kernel, dim3(grid_for((uint32_t)ranges.size(), 4, 4096)), dim3(256), 0, c->stream, d_codes,
                           (const ScanRange *)c->d_ranges.p, (uint32_t)ranges.size(), k, (uint64_t *)keys0.p, (uint32_t *)vals0.p);
    }
    // radix sort of (k-mer, position); the reference's raduls MSD sort / std::sort (agc_compressor.cpp:482-491)
    rocprim::double_buffer<uint64_t> dk((uint64_t *)keys0.p + lo, (uint64_t *)keys1.p + lo);
    rocprim::double_buffer<uint32_t> dv((uint32_t *)vals0.p + lo, (uint32_t *)vals1.p + lo);
    size_t tmp_bytes = 0;
    if (rocprim::radix_sort_pairs(nullptr, tmp_bytes, dk, dv, (size_t)total, 0, 64, c->stream) != hipSuccess)
        return fail(AGC_HIP_ENODEV);
    if ((rc = ensure(c, tmp, tmp_bytes + 256)))
        return fail(rc);
    {
        KTimer t(c, AGC_HIP_K_INDEX);
        if (rocprim::radix_sort_pairs(tmp.p, tmp_bytes, dk, dv, (size_t)total, 0, 64, c->stream) != hipSuccess)
            return fail(AGC_HIP_ENODEV);
    }
    const uint64_t *sk = dk.current();
    const uint32_t *sv = dv.current();
    hipLaunchKernelGGL(singleton_bitmap_kernel, dim3(4096), dim3(256), 0, c->stream, sk, sv, total, (uint32_t *)bitmap.p);
    if (hipGetLastError() != hipSuccess)
        return fail(AGC_HIP_ENODEV);
    std::vector<uint32_t> bm(hi / 32 + 2);
    if (hipMemcpyAsync(bm.data(), bitmap.p, bm.size() * 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess)
        return fail(AGC_HIP_ENODEV);

    // adaptive mode keeps the sorted k-mers (singletons / duplicates are split by the caller)
    if (h_sorted_kmers && h_n_sorted) {
        // valid keys precede the ~0 sentinels; their count = total - (# sentinels): find by binary search on the device copy
        std::vector<uint64_t> probe(1);
        uint64_t a = 0, b = total;
        while (a < b) {
            const uint64_t m = (a + b) / 2;
            if (hipMemcpy(probe.data(), sk + m, 8, hipMemcpyDeviceToHost) != hipSuccess)
                return fail(AGC_HIP_ENODEV);
            if (probe[0] == ~0ULL)
                b = m;
            else
                a = m + 1;
        }
        *h_n_sorted = a;
        if (a > sets_cap) {
            release();
            return AGC_HIP_ECAP;
        }
        if (a && hipMemcpy(h_sorted_kmers, sk, a * 8, hipMemcpyDeviceToHost) != hipSuccess)
            return fail(AGC_HIP_ENODEV);
    }

    // find_splitters_in_contig (agc_compressor.cpp:762-825) over the singleton bitmap
    auto next_set = [&](uint64_t from, uint64_t to) -> uint64_t { // first set bit in [from, to) or ~0
        if (from >= to)
            return ~0ULL;
        uint64_t w = from >> 5;
        uint32_t x = bm[w] & (~0u << (from & 31));
        for (;;) {
            if (x) {
                const uint64_t p = (w << 5) + (uint64_t)__builtin_ctz(x);
                return p < to ? p : ~0ULL;
            }
            if (((++w) << 5) >= to)
                return ~0ULL;
            x = bm[w];
        }
    };
    auto last_set = [&](uint64_t from, uint64_t to) -> uint64_t { // last set bit in [from, to) or ~0
        if (from >= to)
            return ~0ULL;
        uint64_t w = (to - 1) >> 5;
        uint32_t x = bm[w] & (((to - 1) & 31) == 31 ? ~0u : ((1u << (((to - 1) & 31) + 1)) - 1u));
        for (;;) {
            if (x) {
                const uint64_t p = (w << 5) + 31 - (uint64_t)__builtin_clz(x);
                return p >= from ? p : ~0ULL;
            }
            if (w == 0 || (w << 5) <= from)
                return ~0ULL;
            x = bm[--w];
        }
    };
    std::vector<uint64_t> spl_pos;
    for (uint32_t ci = 0; ci < n_ctg; ++ci) {
        const uint64_t b = h_ctg_off[ci], e = h_ctg_off[ci + 1];
        if (e - b < k)
            continue;
        // current_len starts at segment_size: the first singleton k-mer of the contig qualifies; after a
        // splitter ending at p the next check passes at symbol p + segment_size or later
        uint64_t from = b, recent_from = b;
        for (;;) {
            const uint64_t p = next_set(from, e);
            if (p == ~0ULL)
                break;
            spl_pos.push_back(p);
            from = p + segment_size;
            recent_from = p + 1;
        }
        // right-most singleton among the k-mers seen since the last splitter (the k-mer restarts there:
        // candidates end at recent_from + k - 1 or later)
        const uint64_t tail = last_set(recent_from == b ? b : recent_from + k - 1, e);
        if (tail != ~0ULL)
            spl_pos.push_back(tail);
    }
    std::vector<uint64_t> spl(spl_pos.size());
    if (!spl_pos.empty()) {
        DevBuf dpos, dout;
        if ((rc = ensure(c, dpos, spl_pos.size() * 8)) || (rc = ensure(c, dout, spl_pos.size() * 8))) {
            if (dpos.p)
                (void)hipFree(dpos.p);
            return fail(rc);
        }
        (void)hipMemcpyAsync(dpos.p, spl_pos.data(), spl_pos.size() * 8, hipMemcpyHostToDevice, c->stream);
        hipLaunchKernelGGL(kmers_at_kernel, dim3((uint32_t)((spl_pos.size() + 255) / 256)), dim3(256), 0, c->stream, d_codes,
                           (const uint64_t *)dpos.p, (uint32_t)spl_pos.size(), k, (uint64_t *)dout.p);
        (void)hipMemcpyAsync(spl.data(), dout.p, spl.size() * 8, hipMemcpyDeviceToHost, c->stream);
        const hipError_t e2 = hipStreamSynchronize(c->stream);
        (void)hipFree(dpos.p);
        (void)hipFree(dout.p);
        if (e2 != hipSuccess)
            return fail(AGC_HIP_ENODEV);
    }
    release();
    std::sort(spl.begin(), spl.end());
    spl.erase(std::unique(spl.begin(), spl.end()), spl.end());
    *h_n_splitters = spl.size();
    if (spl.size() > cap)
        return AGC_HIP_ECAP;
    if (!spl.empty())
        memcpy(h_splitters, spl.data(), spl.size() * 8);
    return AGC_HIP_OK;
}
