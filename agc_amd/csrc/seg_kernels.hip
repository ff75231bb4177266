// seg_kernels.hip -- from splitter hits to classified segments on the device (gfx950, wave64).
//
// What the reference's workers do per contig on the host, restated for a whole sample in HBM:
//   * the "reset the k-mer after a hit" rule of compress_contig (src/core/agc_compressor.cpp:2007-2036): of the raw hits the
//     scan kernel reports, a hit is taken when it ends at least k symbols after the previous TAKEN hit of its contig;
//   * the cut into segments [split_pos, pos] with their front / back k-mers, and the tail segment (:2018-2048);
//   * the first decision of add_segment (:1275-1330): the key (min, max) of the two splitters' canonical k-mers, its
//     orientation, and the look-up in map_segments (src/core/agc_compressor.h:628) -- here an open-addressing table in HBM
//     whose bucket ranges the look-up kernel stages in LDS;
//   * for every segment whose key is in the table: the descriptor of its LZ encode (SegDesc), longest first, so that the
//     encode kernel can start without the host having seen a single segment.
// The host gets the finished segment table (agc_hip_segment) and goes on with the few segments that need more than a table
// look-up (one splitter: Estimate; destroyed middle splitter: cost vectors).  Integer work on a few ten thousand records: the
// kernels are small; what they buy is that nothing between the scan and the encode launch waits for the host.
#include "dev_common.h"

namespace agc {

// one slot of the (k1, k2) -> group table: the host's PkMap slot (compressor_impl.h), mirrored (include/agc_hip.h: agc_hip_group_slot)
struct GroupSlot {
    uint64_t k1, k2;
    int32_t gid;
    uint32_t used;
};

__host__ __device__ inline uint64_t group_hash(uint64_t k1, uint64_t k2)
{
    uint64_t h = k1 * 0x9E3779B97F4A7C15ULL;
    h ^= (h >> 32) ^ (k2 * 0xC2B2AE3D27D4EB4FULL);
    return h ^ (h >> 29);
}

// what the host receives per segment (include/agc_hip.h: agc_hip_segment)
struct DevSeg {
    uint64_t start; // relative to the contig
    uint64_t front_dir, front_rc, back_dir, back_rc;
    uint32_t ctg, len;
    int32_t map_gid;
    uint8_t front_full, back_full, store_rc, encoded;
};

struct SegCounts { // device counters of one call
    uint32_t n_acc, n_segs, n_known, pad;
    unsigned long long enc_cap; // bytes of encode scratch the known segments take
};

// ---- small device-wide primitives.  The lists here hold a few ten thousand entries: one block scans them in microseconds, and
// a library call (size query + launch, each asking the runtime for the device's properties) costs more host time than that.
__device__ __forceinline__ uint32_t shfl_up_t(uint32_t v, uint32_t d) { return __shfl_up(v, d); }
__device__ __forceinline__ unsigned long long shfl_up_t(unsigned long long v, uint32_t d)
{
    const uint32_t lo = __shfl_up((uint32_t)v, d), hi = __shfl_up((uint32_t)(v >> 32), d);
    return ((unsigned long long)hi << 32) | lo;
}

// out[i] = in[0] + ... + in[i - 1] for i < n (one block of 1024 threads, 8 entries per thread and trip)
template <typename T> __global__ void __launch_bounds__(1024) scan_excl_kernel(const T *__restrict__ in, T *__restrict__ out, uint32_t n)
{
    constexpr uint32_t PER = 8;
    __shared__ T wave_tot[16];
    __shared__ T carry;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0)
        carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024 * PER) {
        const uint32_t i0 = base + threadIdx.x * PER;
        T v[PER], tsum = 0;
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) {
            v[j] = i0 + j < n ? in[i0 + j] : (T)0;
            tsum += v[j];
        }
        T x = tsum; // inclusive scan over the wave
        for (uint32_t o = 1; o < 64; o <<= 1) {
            const T y = shfl_up_t(x, o);
            if (lane >= o)
                x += y;
        }
        if (lane == 63)
            wave_tot[w] = x;
        __syncthreads();
        if (w == 0) {
            const T t = lane < 16 ? wave_tot[lane] : (T)0;
            T e = t;
            for (uint32_t o = 1; o < 16; o <<= 1) {
                const T y = shfl_up_t(e, o);
                if (lane >= o)
                    e += y;
            }
            if (lane < 16)
                wave_tot[lane] = e - t; // exclusive
        }
        __syncthreads();
        T run = carry + wave_tot[w] + (x - tsum);
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) {
            if (i0 + j < n)
                out[i0 + j] = run;
            run += v[j];
        }
        __syncthreads();
        if (threadIdx.x == 1023)
            carry = run;
        __syncthreads();
    }
}

// ---- 1. the hits in position order.  The scan kernel appends hits as its waves find them.  Positions of splitter hits spread
// over the sample, so a counting sort into about n / 2 position buckets leaves a handful per bucket, which one thread orders:
// count -> exclusive scan -> scatter -> per-bucket insertion sort (a degenerate input -- every hit in a few buckets -- is slower,
// never wrong).
__global__ void __launch_bounds__(256) hb_count_kernel(const ScanHit *__restrict__ hits, uint32_t n, uint64_t base, uint32_t sh, uint32_t *__restrict__ cnt)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n)
        atomicAdd(&cnt[(hits[j].pos - base) >> sh], 1u);
}

__global__ void __launch_bounds__(256) hb_scatter_kernel(const ScanHit *__restrict__ hits, uint32_t n, uint64_t base, uint32_t sh, const uint32_t *__restrict__ bstart,
                                                         uint32_t *__restrict__ cur, uint64_t *__restrict__ pos, uint32_t *__restrict__ order)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) {
        const uint64_t p = hits[j].pos;
        const uint32_t b = (uint32_t)((p - base) >> sh);
        const uint32_t slot = bstart[b] + atomicAdd(&cur[b], 1u);
        pos[slot] = p;
        order[slot] = j;
    }
}

__global__ void __launch_bounds__(256) hb_sort_kernel(const uint32_t *__restrict__ bstart, uint32_t n_buckets, uint64_t *__restrict__ pos, uint32_t *__restrict__ order)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_buckets)
        return;
    const uint32_t lo = bstart[b], hi = bstart[b + 1];
    for (uint32_t i = lo + 1; i < hi; ++i) {
        const uint64_t p = pos[i];
        const uint32_t o = order[i];
        uint32_t t = i;
        while (t > lo && pos[t - 1] > p) {
            pos[t] = pos[t - 1];
            order[t] = order[t - 1];
            --t;
        }
        pos[t] = p;
        order[t] = o;
    }
}

// contig of a buffer position (ctg_off: n_ctg + 1 ascending offsets)
__device__ __forceinline__ uint32_t contig_of(const uint64_t *__restrict__ ctg_off, uint32_t n_ctg, uint64_t pos)
{
    uint32_t lo = 0, hi = n_ctg; // ctg_off[lo] <= pos < ctg_off[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (ctg_off[mid] <= pos)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

// ---- 2. contig of every hit (sorted order) ----------------------------------------------------------------------------
__global__ void __launch_bounds__(256) hit_contig_kernel(const uint64_t *__restrict__ pos, uint32_t n, const uint64_t *__restrict__ ctg_off, uint32_t n_ctg,
                                                         uint32_t *__restrict__ ctg)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n)
        ctg[j] = contig_of(ctg_off, n_ctg, pos[j]);
}

// ---- 3. which hits are taken.  A hit whose predecessor (in its contig) ends at least k symbols earlier is taken whatever
// happened before it; it starts a chain, and the thread that owns a chain start resolves the (short) run of closer hits behind it
// in order.  take[n] = 0 closes the array for the exclusive scan that ranks the taken hits (acc_rank[n] = their number).
__global__ void __launch_bounds__(256) hit_accept_kernel(const uint64_t *__restrict__ pos, const uint32_t *__restrict__ ctg, uint32_t n, uint32_t k,
                                                         uint32_t *__restrict__ take)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j == n)
        take[n] = 0;
    if (j >= n)
        return;
    auto is_start = [&](uint32_t t) { return t == 0 || ctg[t] != ctg[t - 1] || pos[t] - pos[t - 1] >= k; };
    if (!is_start(j))
        return;
    uint64_t last = pos[j];
    take[j] = 1;
    for (uint32_t t = j + 1; t < n && !is_start(t); ++t) {
        if (pos[t] >= last + k) {
            take[t] = 1;
            last = pos[t];
        } else
            take[t] = 0;
    }
}

// the list of taken hits: acc_idx[a] = sorted index of the a-th taken hit
__global__ void __launch_bounds__(256) hit_compact_kernel(const uint32_t *__restrict__ take, const uint32_t *__restrict__ acc_rank, uint32_t n,
                                                          uint32_t *__restrict__ acc_idx)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n && take[j])
        acc_idx[acc_rank[j]] = j;
}

// ---- 4. per contig: its taken hits [hits_before[c], hits_before[c + 1]) (the contigs lie in position order, so this is a binary
// search in the list of taken hits), whether it ends in a tail segment, and the exclusive sum of the tails (one block)
__global__ void __launch_bounds__(1024) contig_sums_kernel(const uint64_t *__restrict__ ctg_off, uint32_t n_ctg, uint32_t k, const uint64_t *__restrict__ pos,
                                                           const uint32_t *__restrict__ ctg, const uint32_t *__restrict__ acc_idx,
                                                           const uint32_t *__restrict__ acc_rank, uint32_t n, uint32_t *__restrict__ hits_before,
                                                           uint32_t *__restrict__ tails_before, SegCounts *__restrict__ counts)
{
    __shared__ uint32_t s_b[1024];
    __shared__ uint32_t carry_b;
    const uint32_t n_acc = n ? acc_rank[n] : 0;
    if (threadIdx.x == 0)
        carry_b = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_ctg; base += 1024) {
        const uint32_t c = base + threadIdx.x;
        uint32_t b = 0;
        if (c < n_ctg) {
            // first taken hit whose contig is >= c / > c
            auto lower = [&](uint32_t cc) {
                uint32_t lo = 0, hi = n_acc;
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (ctg[acc_idx[mid]] < cc)
                        lo = mid + 1;
                    else
                        hi = mid;
                }
                return lo;
            };
            const uint32_t h0 = lower(c), h1 = lower(c + 1);
            hits_before[c] = h0;
            if (c + 1 == n_ctg)
                hits_before[n_ctg] = h1;
            const uint64_t len = ctg_off[c + 1] - ctg_off[c];
            // split_pos after the last taken hit (relative to the contig): pos + 1 - k; a tail segment follows when it is < len
            const uint64_t split = h1 > h0 ? pos[acc_idx[h1 - 1]] - ctg_off[c] + 1 - k : 0;
            b = split < len ? 1u : 0u;
        }
        s_b[threadIdx.x] = b;
        __syncthreads();
        for (uint32_t o = 1; o < 1024; o <<= 1) {
            uint32_t vb = 0;
            if (threadIdx.x >= o)
                vb = s_b[threadIdx.x - o];
            __syncthreads();
            s_b[threadIdx.x] += vb;
            __syncthreads();
        }
        if (c < n_ctg)
            tails_before[c] = carry_b + s_b[threadIdx.x] - b;
        __syncthreads();
        if (threadIdx.x == 1023)
            carry_b += s_b[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        tails_before[n_ctg] = carry_b;
        counts->n_acc = n_acc;
        counts->n_segs = n_acc + carry_b;
    }
}

// ---- 5. the segments.  Segment index of taken hit a (a-th taken overall) = a + tails_before[its contig]; the tail segment of
// contig c sits right after the contig's last hit segment.  First the tail segments as far as the contig alone defines them.
__global__ void __launch_bounds__(256) seg_tail_kernel(const uint64_t *__restrict__ ctg_off, uint32_t n_ctg, const uint32_t *__restrict__ hits_before,
                                                       const uint32_t *__restrict__ tails_before, DevSeg *__restrict__ segs)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n_ctg && tails_before[c + 1] != tails_before[c]) {
        DevSeg s;
        s.ctg = c;
        s.front_dir = s.front_rc = s.back_dir = s.back_rc = 0;
        s.front_full = 0;
        s.back_full = 0;
        s.store_rc = 0;
        s.encoded = 0;
        s.map_gid = -1;
        s.start = 0; // (a contig with hits: seg_fill_kernel moves the start behind its last taken hit)
        s.len = (uint32_t)(ctg_off[c + 1] - ctg_off[c]);
        segs[hits_before[c + 1] + tails_before[c]] = s;
    }
}

// one thread per taken hit a: its segment, and -- for the last hit of a contig -- the
// start / front k-mer of the contig's tail segment
__global__ void __launch_bounds__(256) seg_fill_kernel(const ScanHit *__restrict__ hits, const uint32_t *__restrict__ order,
                                                       const uint32_t *__restrict__ ctg, const uint32_t *__restrict__ acc_sorted_idx, const SegCounts *__restrict__ counts,
                                                       uint32_t k, const uint64_t *__restrict__ ctg_off, const uint32_t *__restrict__ hits_before,
                                                       const uint32_t *__restrict__ tails_before, DevSeg *__restrict__ segs)
{
    const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= counts->n_acc)
        return;
    const uint32_t j = acc_sorted_idx[a];
    const uint32_t c = ctg[j];
    const ScanHit h = hits[order[j]];
    const uint32_t t = a - hits_before[c]; // index among the contig's taken hits
    DevSeg s;
    s.ctg = c;
    s.back_dir = h.dir;
    s.back_rc = h.rc;
    s.back_full = 1;
    s.store_rc = 0;
    s.encoded = 0;
    s.map_gid = -1;
    const uint64_t rel = h.pos - ctg_off[c];
    if (t == 0) {
        s.start = 0;
        s.front_dir = s.front_rc = 0;
        s.front_full = 0;
    } else {
        const ScanHit p = hits[order[acc_sorted_idx[a - 1]]];
        s.start = p.pos - ctg_off[c] + 1 - k;
        s.front_dir = p.dir;
        s.front_rc = p.rc;
        s.front_full = 1;
    }
    s.len = (uint32_t)(rel + 1 - s.start);
    segs[a + tails_before[c]] = s;
    // last taken hit of its contig and a tail follows: the tail starts at this hit's split position with this hit's k-mer in front
    if (a + 1 == hits_before[c + 1] && tails_before[c + 1] != tails_before[c]) {
        DevSeg &tl = segs[hits_before[c + 1] + tails_before[c]];
        const uint64_t split = rel + 1 - k;
        tl.start = split;
        tl.len = (uint32_t)(ctg_off[c + 1] - ctg_off[c] - split);
        tl.front_dir = h.dir;
        tl.front_rc = h.rc;
        tl.front_full = 1;
    }
}

// ---- 6. the look-up: (min, max) of the two canonical k-mers -> group.  The table is an open-addressing array (linear probing)
// that a block stages one bucket range of in LDS; it answers the segments whose home slot lies in its range, from LDS as long as
// the probe chain stays inside the range and from HBM beyond it.  grid.x = number of ranges.
constexpr uint32_t GM_RANGE_SLOTS = 4096; // 96 KiB of LDS per block

__device__ __forceinline__ void seg_key(const DevSeg &s, uint64_t &k1, uint64_t &k2, bool &rc)
{
    // canonical k-mers (CKmer::data, kmer.h:350-357) and the orientation rule of add_segment (agc_compressor.cpp:1286-1301)
    const uint64_t f = s.front_dir < s.front_rc ? s.front_dir : s.front_rc, b = s.back_dir < s.back_rc ? s.back_dir : s.back_rc;
    if (f < b) {
        k1 = f;
        k2 = b;
        rc = false;
    } else {
        k1 = b;
        k2 = f;
        rc = true;
    }
}

__global__ void __launch_bounds__(1024) group_lookup_kernel(const GroupSlot *__restrict__ table, uint64_t mask, DevSeg *__restrict__ segs,
                                                            const SegCounts *__restrict__ counts, uint32_t staged)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    GroupSlot *s_tab = (GroupSlot *)s_raw;
    const uint64_t n_slots = mask + 1;
    const uint64_t range_slots = n_slots < GM_RANGE_SLOTS ? n_slots : GM_RANGE_SLOTS;
    const uint64_t r0 = (uint64_t)blockIdx.x * range_slots;
    if (staged) {
        const uint4 *src = (const uint4 *)(table + r0);
        uint4 *dst = (uint4 *)s_tab;
        for (uint32_t t = threadIdx.x; t < range_slots * sizeof(GroupSlot) / 16; t += blockDim.x)
            dst[t] = src[t];
        __syncthreads();
    }
    const uint32_t n = counts->n_segs;
    const uint32_t stride = staged ? blockDim.x : blockDim.x * gridDim.x;
    for (uint32_t si = staged ? threadIdx.x : blockIdx.x * blockDim.x + threadIdx.x; si < n; si += stride) {
        DevSeg s = segs[si];
        if (!s.front_full || !s.back_full)
            continue;
        uint64_t k1, k2;
        bool rc;
        seg_key(s, k1, k2, rc);
        uint64_t i = group_hash(k1, k2) & mask;
        if (staged && i / range_slots != blockIdx.x)
            continue; // another block's range
        int32_t gid = -1;
        for (;;) {
            const GroupSlot g = (staged && i - r0 < range_slots) ? s_tab[i - r0] : table[i];
            if (!g.used)
                break;
            if (g.k1 == k1 && g.k2 == k2) {
                gid = g.gid;
                break;
            }
            i = (i + 1) & mask;
        }
        segs[si].map_gid = gid;
        segs[si].store_rc = rc ? 1 : 0;
    }
}

// ---- 7. encode descriptors of the segments whose group is known (and has its reference in HBM).  flag / cap hold n_ub + 1
// entries (the last one 0) so that their exclusive scans end in the totals.
__global__ void __launch_bounds__(256) known_flag_kernel(const DevSeg *__restrict__ segs, const SegCounts *__restrict__ counts, const RefDesc *__restrict__ refs,
                                                         uint32_t n_refs, uint32_t n_ub, uint32_t *__restrict__ flag, unsigned long long *__restrict__ cap)
{
    const uint32_t si = blockIdx.x * blockDim.x + threadIdx.x;
    if (si > n_ub)
        return;
    uint32_t f = 0;
    unsigned long long cp = 0;
    if (si < n_ub && si < counts->n_segs) {
        const DevSeg s = segs[si];
        if (s.front_full && s.back_full && s.map_gid >= 16 && (uint32_t)s.map_gid < n_refs && refs[s.map_gid].valid) {
            f = 1;
            cp = (((unsigned long long)s.len + 5ULL * s.len / 16 + 64) + 15) & ~15ULL; // as prepare_batch sizes a delta's slot
        }
    }
    flag[si] = f;
    cap[si] = cp;
}

__global__ void __launch_bounds__(256) known_emit_kernel(DevSeg *__restrict__ segs, SegCounts *__restrict__ counts, const uint32_t *__restrict__ flag,
                                                         const uint32_t *__restrict__ known_rank, const unsigned long long *__restrict__ cap_off, uint32_t n_ub,
                                                         PackedView pv, const uint64_t *__restrict__ ctg_off, SegDesc *__restrict__ descs)
{
    const uint32_t si = blockIdx.x * blockDim.x + threadIdx.x;
    if (si == 0) { // the totals
        counts->n_known = known_rank[n_ub];
        counts->enc_cap = cap_off[n_ub];
    }
    if (si >= n_ub || !flag[si])
        return;
    const DevSeg s = segs[si];
    const uint32_t r = known_rank[si];
    SegDesc d;
    d.text = {pv.words, pv.esc_index, pv.esc_bytes, ctg_off[s.ctg] + s.start, s.len, (uint32_t)s.store_rc};
    d.maybe = nullptr;
    d.out_off = cap_off[si];
    d.ref_slot = (uint32_t)s.map_gid;
    d.flags = 0;
    d.idx = r;
    d.pad = 0;
    descs[r] = d;
    segs[si].encoded = 1;
}

// descriptors in processing order: longest first, to the bucket of LEN_BUCKET symbols (the order only decides which waves start
// first -- the long segments, so that the short ones fill the tail of the launch; results are indexed by SegDesc::idx)
constexpr uint32_t LEN_BUCKET_SHIFT = 8, LEN_BUCKETS = 8192;
__device__ __forceinline__ uint32_t len_bucket(uint32_t len)
{
    const uint32_t b = len >> LEN_BUCKET_SHIFT;
    return LEN_BUCKETS - 1 - (b < LEN_BUCKETS ? b : LEN_BUCKETS - 1); // bucket 0 = the longest
}

// One atomic per distinct key and wavefront instead of one per lane: the segments of a sample cut at the default segment size fall
// into a handful of length buckets, and 50 k atomics on three addresses serialise (0.54 ms per kernel, measured).  Returns the
// lane's rank among the arrivals at ctr[key].
__device__ __forceinline__ uint32_t wave_agg_add(uint32_t *__restrict__ ctr, uint32_t key, bool active)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t rank = 0;
    uint64_t todo = __ballot(active);
    while (todo) {
        const uint32_t leader = (uint32_t)__builtin_ctzll(todo);
        const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)key, (int)leader);
        const bool mine = active && key == k;
        const uint64_t same = __ballot(mine);
        uint32_t base = 0;
        if (lane == leader)
            base = atomicAdd(&ctr[k], (uint32_t)__builtin_popcountll(same));
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)leader);
        if (mine)
            rank = base + (uint32_t)__builtin_popcountll(same & ((1ULL << lane) - 1ULL));
        todo &= ~same;
    }
    return rank;
}

__global__ void __launch_bounds__(256) known_len_count_kernel(const SegDesc *__restrict__ descs, const SegCounts *__restrict__ counts, uint32_t *__restrict__ cnt)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = r < counts->n_known;
    (void)wave_agg_add(cnt, active ? len_bucket(descs[r].text.len) : 0u, active);
}

__global__ void __launch_bounds__(256) known_order_kernel(const SegDesc *__restrict__ descs, const SegCounts *__restrict__ counts, const uint32_t *__restrict__ bstart,
                                                          uint32_t *__restrict__ cur, SegDesc *__restrict__ out)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = r < counts->n_known;
    SegDesc d;
    uint32_t b = 0;
    if (active) {
        d = descs[r];
        b = len_bucket(d.text.len);
    }
    const uint32_t rank = wave_agg_add(cur, b, active);
    if (active)
        out[bstart[b] + rank] = d;
}

} // namespace agc
