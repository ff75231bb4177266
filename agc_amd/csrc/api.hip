// api.hip -- the C ABI of include/agc_hip.h: context, HBM arenas, kernel launches.
// Host-side logic here is orchestration only (buffer management, sorting sparse hit
// lists, sizing tables with the reference's double arithmetic); all symbol-level work is
// done by the kernels in scan_kernels.hip / lz_kernels.hip.  There is no CPU fallback.
#include "../../include/agc_hip.h"
#include "dev_common.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <future>
#include <mutex>
#include <numeric>
#include <string>
#include <vector>

#include "scan_kernels.hip"
#include "pack_kernels.hip"
#include "lz_kernels.hip"
#include "seg_kernels.hip"
#include "zstd_kernels.hip"

using namespace agc;

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

struct ArenaChunk {
    uint8_t *p;
    size_t size, used;
};

// a 2-bit packed buffer the LZ entry points read sequences from (sym_view.h): the caller's sample, or a temporary one
struct PackedSrc {
    const uint32_t *words = nullptr;
    const int32_t *esc_index = nullptr;
    const uint8_t *esc_bytes = nullptr;
    uint64_t n_symbols = 0;
};

// context-owned packed form of byte input (the entry points that take one byte per symbol pack what they are given first:
// every LZ kernel reads the 2-bit layout only)
struct PackTemp {
    DevBuf words, index, esc, cnt;
};

} // namespace

struct agc_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t zstream = nullptr; // the entropy stage's own stream: agc_hip_zstd17_batch may run beside every other entry point
    hipStream_t zstream2 = nullptr; // ... and a second one: the one-lane kernel (inputs > 16 KiB) runs BESIDE the group kernel
    hipEvent_t zev_a = nullptr, zev_b = nullptr;
    hipEvent_t zev_wait = nullptr; // (blocking-sync: the thread that waits for a launch of the entropy stage sleeps, its core is the host pool's)
    bool zstd_background = false;  // agc_hip_zstd17_background: launches leave the LDS to the kernels of the other streams
    std::string err;

    // splitter set
    std::vector<uint64_t> spl;       // host copy (unique)
    DevBuf d_table, d_bloom, d_bloom2, d_sbloom;
    uint32_t sbloom_k = 0;          // the packed scan's suffix filter: built for this k ...
    size_t sbloom_n = ~(size_t)0;   // ... and this many splitters
    uint64_t table_mask = 0;

    // references
    std::vector<RefDesc> refs;       // indexed by gid
    DevBuf d_refs;
    bool refs_dirty = true;
    size_t refs_dirty_lo = 0, refs_dirty_hi = 0; // groups [lo, hi) changed since the table went to the device (refs_on_dev entries are there)
    size_t refs_on_dev = 0;
    std::vector<ArenaChunk> arena;
    // agc_hip_ref_store_begin_packed / _end: two slots, a stream and buffers of their own (the steps' stream never waits for them)
    struct RefStore {
        DevBuf d_slices, d_lag, d_out;
        hipEvent_t done = nullptr;
        std::atomic<bool> pending{false}; // (_end may run on the caller's bookkeeping thread)
    } ref_store[2];
    hipStream_t ref_store_stream = nullptr;
    // the NEXT chunk, allocated ahead of need by a helper thread (arena_alloc): a hipMalloc of GBs is 30 ms per GB on a box whose
    // VRAM this process touches for the first time -- 150 ms in the middle of a step (profiles/r6/)
    std::future<ArenaChunk> arena_spare;

    // the > 64 KiB dynamic-LDS attribute of a kernel is a property of the device the context runs on: set once per context
    bool lds_scan_set = false, lds_lookup_set = false;

    // scratch
    DevBuf d_esc_jobs, d_flags;
    // the (k1, k2) -> group table (mirror of the host's map_segments) and the work area of agc_hip_segments_packed
    DevBuf d_gmap, d_gmap_stage, d_segwork, d_segtmp;
    uint64_t gmap_slots = 0;
    void *h_gmap_stage = nullptr; // pinned staging of agc_hip_group_map_update
    size_t h_gmap_stage_cap = 0;
    hipEvent_t gmap_ev = nullptr;
    bool gmap_ev_valid = false;
    uint32_t *h_zsizes = nullptr; // pinned: frame sizes of the entropy launch in flight (a pageable destination makes the "async" copy
    size_t h_zsizes_cap = 0;      // wait for the launch inside the call, spinning: a core of the host pool's for the whole launch)
    void *h_segcounts = nullptr; // pinned: SegCounts of the call in flight (+ 64: the count of the encode launched from them)
    // what agc_hip_segments_packed left on the device for agc_hip_segments_encode_known
    struct SegState {
        bool valid = false;
        uint32_t n_ub = 0;
        uint64_t total = 0;
        agc_hip_packed pk{};
        void *segs = nullptr, *counts = nullptr, *d_ctg_off = nullptr;
        void *flag = nullptr, *capv = nullptr, *known_rank = nullptr, *cap_off = nullptr, *descs = nullptr, *lcnt = nullptr;
    } seg_state;
    DevBuf d_ranges, d_hits, d_counter, d_segs, d_slices, d_scratch, d_resv, d_resp, d_dstoff, d_compact,
        d_jobs, d_counts, d_in, d_pp_cnt, d_pp_off, d_pp_total, d_lag, d_sample, d_zsrc, d_zdst, d_zws, d_zjobs, d_zsize, d_zout, d_zdstoff, d_maybe, d_fjobs;

    PackTemp pk1;        // byte-input entry points on the first stream
    PackTemp pk_sample;  // agc_hip_sample_pack
    std::mutex host_alloc_mtx; // host_allocs: agc_hip_host_alloc / _free may be called from the thread that collects an encode
    // pinned staging ring of the small host -> device uploads (descriptors, offsets, tables): a copy from pageable memory makes the
    // calling thread wait for the copy engine -- behind whatever else it is moving, e.g. the 22 MB of a sample's deltas
    uint8_t *up_ring = nullptr;
    size_t up_cap = 0, up_head = 0;
    std::mutex up_mtx;
    // the ring in UP_PARTS parts: a part remembers the streams that copied out of it and, when the head leaves it, an event on each;
    // the head waits for those events when it comes back (a ring's length later: they are long done -- the whole-ring wait for
    // every stream at the wrap cost a step 7 ms once in 16 samples)
    static constexpr int UP_PARTS = 4, UP_STREAMS = 12;
    struct UpPart {
        hipStream_t st[UP_STREAMS] = {};
        hipEvent_t ev[UP_STREAMS] = {};
        int n = 0;         // streams noted since the head came in
        int n_recorded = 0; // events recorded when it left
    } up_part[UP_PARTS];

    // second LZ lane: agc_hip_lz_encode_begin_dev / _end run the encode of a whole sample on `stream2` with their own scratch,
    // beside the estimates / cost vectors / index builds the caller goes on with on `stream`
    struct Lane2 {
        DevBuf d_segs, d_counter, d_resv, d_resp, d_scratch, d_dstoff, d_compact;
        DevBuf d_n; // the lane's own copy of a segment count made on the device (the work area it came from is re-laid-out by the next sample)
        PackTemp pk;
        uint32_t *h_lens = nullptr; // pinned (a device-to-host copy into pageable memory would make begin wait for the kernel)
        size_t h_lens_cap = 0;
        uint32_t n = 0;          // segments of the encode in flight
        const uint32_t *n_pinned = nullptr; // != nullptr: their number arrives with the parse (descriptors made on the device)
        bool pending = false, timed = false;
        hipEvent_t e0 = nullptr, e1 = nullptr, ready = nullptr;
        // `done`: recorded behind the parse of the encode in flight.  A caller may collect that encode from another thread while
        // this one already works on the next sample (agc_hip_lz_encode_end only touches this lane): whatever overwrites a buffer
        // the parse reads -- the sample staging buffers -- waits for the event on its own stream (sample_buffer, prefetch)
        hipEvent_t done = nullptr;
        bool done_valid = false;
        hipStream_t s = nullptr; // the lane's stream
    } l2, l3; // (l3: a second encode in flight -- the segments of a sample whose group was minted by that very sample, launched at
              // commit time behind the whole-sample encode of l2 and collected by the same bookkeeping task)
    hipStream_t stream2 = nullptr, stream3 = nullptr;
    Lane2 &lane(int which) { return which == 2 ? l3 : l2; }

    // the NEXT sample, started ahead of its turn (agc_hip_prefetch_packed_dev): expansion into one of two staging buffers and the
    // packed splitter scan, on a stream of their own with their own scratch -- they fill the gaps the sample in front leaves on the
    // GPU while the host registers its segments
    struct Prefetch {
        hipStream_t stream = nullptr;
        DevBuf d_ranges, d_hits, d_counter;
        uint32_t *h_count = nullptr; // pinned
        const void *words = nullptr; // identity of the packed sample in flight
        uint64_t n_symbols = 0;
        uint32_t dev_cap = 0, k = 0, n_ctg = 0;
        uint64_t first_off = 0, last_off = 0;
        bool valid = false, scanned = false;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        bool timed = false;
    } pf;

    // agc_hip_pack_fasta_begin / _end: raw FASTA bodies -> the packed sample on a stream of its own (pack_kernels.hip)
    struct PackFasta {
        hipStream_t stream = nullptr;
        DevBuf d_state, d_rng, d_off; // the three counters; the ranges; n_ctg + 1 symbol offsets + the total
        DevBuf d_tcnt, d_toff;        // two-pass variant: symbols per tile, symbols in front of every tile
        uint64_t *h_res = nullptr;    // pinned: offsets, total, escaped-block count
        size_t h_res_cap = 0;
        uint32_t n_ctg = 0;
        uint64_t n_raw = 0, esc_cap = 0;
        bool pending = false, timed = false;
        hipEvent_t e0 = nullptr, e1 = nullptr;
    } pfa;

    // the parse in chunks (launch_parse): chunk list, logs and per-chunk output of the first stream and of the two encode lanes
    struct ChunkBufs {
        DevBuf d_jobs, d_seg0, d_logs, d_logn, d_out;
        // the chunk logs of a human-size batch are 0.7 GB: a launch of thousands of parses that is NOT parsed in chunks has a helper
        // thread allocate them, so that the first batch that is -- one text of a few hundred kb among 3 000, twice in twenty samples --
        // does not pay a hipMalloc inside a step (27-41 ms on a box whose VRAM the process touches for the first time)
        std::future<DevBuf> logs_ahead;
        bool logs_asked = false;
    } chunk_bufs[3];

    // pinned host allocations handed out by agc_hip_host_alloc
    std::vector<void *> host_allocs;

    // timing
    bool timing = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, zev0 = nullptr, zev1 = nullptr;
    // KTimer's event pairs: recorded around a launch and READ LATER (when the ring comes round, or by agc_hip_timing_get) -- timing a
    // run must not add a wait behind every kernel of the steps' stream (rounds 1-5 did: hipEventSynchronize in KTimer's destructor)
    struct TimerPair {
        hipEvent_t a = nullptr, b = nullptr;
        int which = -1; // >= 0: recorded, not read yet
    };
    std::vector<TimerPair> tpairs;
    size_t tpair_next = 0;
    double ms[AGC_HIP_K_COUNT] = {0};
    uint64_t launches[AGC_HIP_K_COUNT] = {0};
};

#define HIPCHK(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                        \
            return e_ == hipErrorOutOfMemory ? AGC_HIP_ENOMEM : AGC_HIP_ENODEV;                    \
        }                                                                                          \
    } while (0)

#define CHK(expr)                                                                                  \
    do {                                                                                           \
        int r_ = (expr);                                                                           \
        if (r_ != AGC_HIP_OK)                                                                      \
            return r_;                                                                             \
    } while (0)

namespace {

int ensure(agc_hip_ctx *c, DevBuf &b, size_t bytes, hipStream_t stream = nullptr)
{
    if (bytes <= b.cap)
        return AGC_HIP_OK;
    if (!stream)
        stream = c->stream;
    // headroom: batches of one collection differ by a few percent in size; growing in big steps keeps
    // hipFree/hipMalloc (hundreds of ms for multi-GB buffers) out of the steady state
    size_t want = std::max(bytes + bytes / 4, b.cap + b.cap / 2);
    want = (want + 255) & ~(size_t)255;
    static const bool laps = getenv("AGC_HIP_LAPS") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    const size_t had = b.cap;
    if (b.p) {
        HIPCHK(c, hipStreamSynchronize(stream));
        HIPCHK(c, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    HIPCHK(c, hipMalloc(&b.p, want));
    b.cap = want;
    if (laps && want >= ((size_t)16 << 20))
        fprintf(stderr, "    ensure: a device buffer grows from %.1f to %.1f MB (asked: %.1f) in %.3f ms\n", had / 1e6, want / 1e6, bytes / 1e6,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    return AGC_HIP_OK;
}

size_t arena_next_size(const agc_hip_ctx *c)
{
    // a chunk is at least half of what the arena holds already: the number of hipMalloc calls of a run grows with the logarithm of
    // its references, not with their number
    size_t held = 0;
    for (const ArenaChunk &ch : c->arena)
        held += ch.size;
    return std::max((size_t)256 << 20, std::min(held / 2, (size_t)8 << 30));
}

int arena_alloc(agc_hip_ctx *c, size_t bytes, uint8_t **out)
{
    bytes = (bytes + 255) & ~(size_t)255;
    static const bool laps = getenv("AGC_HIP_LAPS") != nullptr;
    if (c->arena.empty() || c->arena.back().used + bytes > c->arena.back().size) {
        const auto t0 = std::chrono::steady_clock::now();
        ArenaChunk nc{nullptr, 0, 0};
        if (c->arena_spare.valid()) { // (allocated since the current chunk was half full: long done)
            nc = c->arena_spare.get();
            if (nc.p && nc.size < bytes) {
                (void)hipFree(nc.p);
                nc = ArenaChunk{nullptr, 0, 0};
            }
        }
        if (!nc.p) {
            const size_t sz = std::max(bytes, arena_next_size(c));
            uint8_t *p = nullptr;
            HIPCHK(c, hipMalloc((void **)&p, sz + 4096)); // tail slack: 16-byte over-reads never leave the allocation
            nc = ArenaChunk{p, sz, 0};
        }
        c->arena.push_back(nc);
        if (laps)
            fprintf(stderr, "    arena: a chunk of %.1f MB becomes current in %.3f ms\n", nc.size / 1e6,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    ArenaChunk &ch = c->arena.back();
    *out = ch.p + ch.used;
    ch.used += bytes;
    // the chunk after this one is asked for as soon as this one is half full, on a thread of its own
    if (!c->arena_spare.valid() && ch.used * 2 >= ch.size) {
        const size_t sz = arena_next_size(c);
        const int dev = c->device;
        c->arena_spare = std::async(std::launch::async, [sz, dev] {
            uint8_t *p = nullptr;
            if (hipSetDevice(dev) != hipSuccess || hipMalloc((void **)&p, sz + 4096) != hipSuccess)
                return ArenaChunk{nullptr, 0, 0}; // (arena_alloc then allocates when the chunk is needed, and reports)
            return ArenaChunk{p, sz, 0};
        });
    }
    return AGC_HIP_OK;
}

void ktimer_read(agc_hip_ctx *c, agc_hip_ctx::TimerPair &p)
{
    if (p.which < 0)
        return;
    float ms = 0;
    if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
        c->ms[p.which] += ms;
        c->launches[p.which] += 1;
    }
    p.which = -1;
}

void ktimer_read_all(agc_hip_ctx *c)
{
    for (auto &p : c->tpairs)
        ktimer_read(c, p);
}

struct KTimer {
    agc_hip_ctx *c;
    agc_hip_ctx::TimerPair *p = nullptr;
    int which;
    KTimer(agc_hip_ctx *c_, int w) : c(c_), which(w) // (w < 0: not timed -- a launch on the second lane)
    {
        if (!c->timing || which < 0)
            return;
        if (c->tpairs.empty())
            c->tpairs.resize(256);
        p = &c->tpairs[c->tpair_next];
        c->tpair_next = (c->tpair_next + 1) % c->tpairs.size();
        ktimer_read(c, *p); // (a pair recorded 256 launches ago: long done)
        if ((!p->a && hipEventCreate(&p->a) != hipSuccess) || (!p->b && hipEventCreate(&p->b) != hipSuccess)) {
            p = nullptr;
            return;
        }
        (void)hipEventRecord(p->a, c->stream);
    }
    ~KTimer()
    {
        if (p) {
            (void)hipEventRecord(p->b, c->stream);
            p->which = which;
        }
    }
};

// the entropy stage's kernels: own events, own stream (only that call writes ms[AGC_HIP_K_ZSTD])
struct ZTimer {
    agc_hip_ctx *c;
    explicit ZTimer(agc_hip_ctx *c_) : c(c_)
    {
        if (c->timing)
            (void)hipEventRecord(c->zev0, c->zstream);
    }
    ~ZTimer()
    {
        if (c->timing) {
            (void)hipEventRecord(c->zev1, c->zstream);
            (void)hipEventSynchronize(c->zev1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, c->zev0, c->zev1);
            c->ms[AGC_HIP_K_ZSTD] += ms;
            c->launches[AGC_HIP_K_ZSTD] += 1;
        }
    }
};

int ensure_z(agc_hip_ctx *c, DevBuf &b, size_t bytes) { return ensure(c, b, bytes, c->zstream); }

// host -> device, without the calling thread waiting: through the context's pinned ring (a slot is reused a ring's length later:
// tens of steps; the wrap waits for the LZ streams once to be sure).  Large copies go the ordinary way.
int upload(agc_hip_ctx *c, void *d_dst, const void *h_src, size_t bytes, hipStream_t st)
{
    // (AGC_HIP_UPLOAD_RING_MB: a smaller ring for the tests -- the head then comes back to a part, and waits for its events, many
    // times in a small archive; a piece is at most half a part)
    static const size_t RING = [] {
        const char *e = getenv("AGC_HIP_UPLOAD_RING_MB");
        const long mb = e ? atol(e) : 64;
        return (size_t)(mb < 1 ? 1 : mb > 64 ? 64 : mb) << 20;
    }();
    const size_t PART = RING / agc_hip_ctx::UP_PARTS, MAX_PIECE = std::min<size_t>((size_t)8 << 20, PART / 2);
    if (!bytes)
        return AGC_HIP_OK;
    if (bytes > MAX_PIECE) {
        HIPCHK(c, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, st));
        return AGC_HIP_OK;
    }
    // (the lock is held until the copy is queued: the event a part records on a stream when the head leaves is then behind every
    // copy out of that part)
    std::lock_guard<std::mutex> lk(c->up_mtx);
    if (!c->up_ring) {
        HIPCHK(c, hipHostMalloc((void **)&c->up_ring, RING, hipHostMallocDefault));
        c->up_cap = RING;
    }
    const size_t need = (bytes + 255) & ~(size_t)255;
    size_t part = c->up_head ? (c->up_head - 1) / PART : 0; // (a head at a part's very end still belongs to it)
    if (c->up_head + need > (part + 1) * PART) {
        agc_hip_ctx::UpPart &old_part = c->up_part[part];
        for (int i = 0; i < old_part.n; ++i) {
            if (!old_part.ev[i])
                HIPCHK(c, hipEventCreateWithFlags(&old_part.ev[i], hipEventDisableTiming));
            HIPCHK(c, hipEventRecord(old_part.ev[i], old_part.st[i]));
        }
        old_part.n_recorded = old_part.n;
        old_part.n = 0;
        part = (part + 1) % agc_hip_ctx::UP_PARTS;
        c->up_head = part * PART;
        agc_hip_ctx::UpPart &new_part = c->up_part[part];
        for (int i = 0; i < new_part.n_recorded; ++i)
            HIPCHK(c, hipEventSynchronize(new_part.ev[i]));
        new_part.n_recorded = 0;
    }
    agc_hip_ctx::UpPart &P = c->up_part[part];
    int k = 0;
    while (k < P.n && P.st[k] != st)
        ++k;
    if (k == P.n) {
        if (P.n == agc_hip_ctx::UP_STREAMS) { // (more streams than a part has room for: make room the slow way)
            HIPCHK(c, hipStreamSynchronize(P.st[0]));
            P.st[0] = st;
        } else
            P.st[P.n++] = st;
    }
    uint8_t *slot = c->up_ring + c->up_head;
    c->up_head += need;
    std::memcpy(slot, h_src, bytes);
    HIPCHK(c, hipMemcpyAsync(d_dst, slot, bytes, hipMemcpyHostToDevice, st));
    return AGC_HIP_OK;
}

// the entropy stage's stream yields to the streams of the steps wherever the hardware queues let it
hipError_t create_low_priority_stream(hipStream_t *s)
{
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess)
        return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
    return hipStreamCreateWithPriority(s, hipStreamNonBlocking, lo);
}

int upload_refs(agc_hip_ctx *c)
{
    if (!c->refs_dirty)
        return AGC_HIP_OK;
    // Only what changed goes over: a registration adds descriptors (a group gets its reference once, agc_hip_ref_register* refuses a
    // second), it never changes one a parse in flight may be reading -- so no lane is waited for and a step moves some KB instead of
    // the whole table (2.8 MB for the 50 k groups of a human sample: 0.3 ms of the driving thread and a twentieth of the pinned ring
    // per sample; the wait for the whole-sample encode on lane 0 was 7 ms of the first step after an idle pipeline).  A table that has
    // to move to a larger buffer is the exception: lanes first, then everything.
    size_t lo = c->refs_dirty_lo, hi = std::min(c->refs_dirty_hi, c->refs.size());
    const size_t need = std::max<size_t>(1, c->refs.size()) * sizeof(RefDesc);
    if (need > c->d_refs.cap || c->refs_on_dev == 0) {
        for (auto *ln : {&c->l2, &c->l3})
            if (ln->pending)
                HIPCHK(c, hipStreamSynchronize(ln->s)); // the encode in flight reads the buffer that is about to be freed
        CHK(ensure(c, c->d_refs, need));
        lo = 0;
        hi = c->refs.size();
    }
    if (hi > lo) {
        CHK(upload(c, (RefDesc *)c->d_refs.p + lo, c->refs.data() + lo, (hi - lo) * sizeof(RefDesc), c->stream));
        if ((hi - lo) * sizeof(RefDesc) > ((size_t)8 << 20))
            HIPCHK(c, hipStreamSynchronize(c->stream)); // (beyond upload()'s staging: the vector may be reallocated by the next register)
    }
    c->refs_on_dev = c->refs.size();
    c->refs_dirty = false;
    c->refs_dirty_lo = c->refs_dirty_hi = 0;
    return AGC_HIP_OK;
}

uint32_t grid_for(uint32_t n_items, uint32_t per_block, uint32_t max_blocks)
{
    uint64_t b = ((uint64_t)n_items + per_block - 1) / per_block;
    if (b < 1)
        b = 1;
    return (uint32_t)std::min<uint64_t>(b, max_blocks);
}

// bytes (one per symbol, device) -> 2-bit layout in the context-owned buffers of `t`, queued on `st`; every block may be
// escaped (room for all of them), so nothing has to be read back
int pack_bytes(agc_hip_ctx *c, const uint8_t *d_codes, uint64_t n, PackTemp &t, hipStream_t st, PackedSrc &out)
{
    out = PackedSrc();
    if (!n)
        return AGC_HIP_OK;
    const uint64_t n_blocks = (n + PACK_BLOCK - 1) / PACK_BLOCK;
    CHK(ensure(c, t.words, n_blocks * (PACK_BLOCK / 4) + 64, st));
    CHK(ensure(c, t.index, n_blocks * 4 + 64, st));
    CHK(ensure(c, t.esc, n_blocks * PACK_BLOCK + 64, st));
    CHK(ensure(c, t.cnt, 64, st));
    HIPCHK(c, hipMemsetAsync(t.cnt.p, 0, 4, st));
    hipLaunchKernelGGL(pack_codes_kernel, dim3((uint32_t)std::min<uint64_t>((n_blocks + 3) / 4, 65536)), dim3(256), 0, st, d_codes, n, (uint32_t *)t.words.p,
                       (int32_t *)t.index.p, (uint8_t *)t.esc.p, (uint32_t *)t.cnt.p, (uint32_t)std::min<uint64_t>(n_blocks, 0x7fffffffu));
    HIPCHK(c, hipGetLastError());
    out.words = (const uint32_t *)t.words.p;
    out.esc_index = (const int32_t *)t.index.p;
    out.esc_bytes = (const uint8_t *)t.esc.p;
    out.n_symbols = n;
    return AGC_HIP_OK;
}

// the sequences [h_off[i], h_off[i] + h_len[i]) of a byte buffer: the range they span is packed, off2 = their offsets in it
int pack_range(agc_hip_ctx *c, const uint8_t *d_base, uint32_t n, const uint64_t *h_off, const uint32_t *h_len, PackTemp &t, hipStream_t st,
               PackedSrc &out, std::vector<uint64_t> &off2)
{
    uint64_t lo = ~0ULL, hi = 0;
    for (uint32_t i = 0; i < n; ++i)
        if (h_len[i]) {
            lo = std::min<uint64_t>(lo, h_off[i]);
            hi = std::max<uint64_t>(hi, h_off[i] + h_len[i]);
        }
    off2.assign(n, 0);
    if (hi <= lo) {
        out = PackedSrc();
        return AGC_HIP_OK;
    }
    for (uint32_t i = 0; i < n; ++i)
        off2[i] = h_len[i] ? h_off[i] - lo : 0;
    return pack_bytes(c, d_base + lo, hi - lo, t, st, out);
}

PackedSrc src_of(const agc_hip_packed *pk) 
{
    PackedSrc s;
    s.words = pk->d_words;
    s.esc_index = pk->d_esc_index;
    s.esc_bytes = pk->d_esc_bytes;
    s.n_symbols = pk->n_symbols;
    return s;
}

bool slices_inside(const PackedSrc &src, uint32_t n, const uint64_t *h_off, const uint32_t *h_len)
{
    for (uint32_t i = 0; i < n; ++i)
        if (h_len[i] && (h_off[i] > src.n_symbols || h_len[i] > src.n_symbols - h_off[i]))
            return false;
    return true;
}

} // namespace

// ===========================================================================
extern "C" {

uint32_t agc_hip_abi_version(void) { return 2; }

int agc_hip_create(agc_hip_ctx **out, int device)
{
    if (!out)
        return AGC_HIP_EINVAL;
    *out = nullptr;
    // The ROCm runtime maps the HIP streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4).  A context has eight
    // streams (steps, two encode lanes, scan prefetch, FASTA pack, two entropy streams + the caller's): on four queues they share, and a
    // wait on an idle stream lasts until the kernels of the stream it shares a queue with are done (measured: pack_fasta_end's
    // hipStreamSynchronize 1 ms behind the followers' encode, profiles/r6/).  Only effective when this is the process's first HIP call
    // (the CLI); a Python caller that initialises HIP first sets the variable itself (bench.py does).
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n)
        return AGC_HIP_ENODEV;
    agc_hip_ctx *c = new agc_hip_ctx();
    c->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        create_low_priority_stream(&c->zstream) != hipSuccess || create_low_priority_stream(&c->zstream2) != hipSuccess ||
        hipEventCreateWithFlags(&c->zev_a, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->zev_b, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->zev_wait, hipEventDisableTiming | hipEventBlockingSync) != hipSuccess ||
        // (the second lane at the first stream's priority: measured with a low-priority lane the whole-sample encode is starved by
        // the classification kernels until they are done and ends up on the critical path of the NEXT sample's launch)
        hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&c->l2.e0) != hipSuccess ||
        hipEventCreate(&c->l2.e1) != hipSuccess || hipEventCreateWithFlags(&c->l2.ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->l2.done, hipEventDisableTiming) != hipSuccess ||
        hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&c->l3.e0) != hipSuccess ||
        hipEventCreate(&c->l3.e1) != hipSuccess || hipEventCreateWithFlags(&c->l3.ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->l3.done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess ||
        hipEventCreate(&c->ev1) != hipSuccess || hipEventCreate(&c->zev0) != hipSuccess || hipEventCreate(&c->zev1) != hipSuccess) {
        delete c;
        return AGC_HIP_ENODEV;
    }
    c->l2.s = c->stream2;
    c->l3.s = c->stream3;
    *out = c;
    return AGC_HIP_OK;
}

void agc_hip_destroy(agc_hip_ctx *c)
{
    if (!c)
        return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->zstream)
        (void)hipStreamSynchronize(c->zstream);
    if (c->zstream2)
        (void)hipStreamSynchronize(c->zstream2);
    for (hipStream_t s_ : {c->stream2, c->stream3})
        if (s_)
            (void)hipStreamSynchronize(s_);
    for (auto &cb : c->chunk_bufs)
        for (DevBuf *b : {&cb.d_jobs, &cb.d_seg0, &cb.d_logs, &cb.d_logn, &cb.d_out})
            if (b->p)
                (void)hipFree(b->p);
    if (c->pfa.stream) {
        (void)hipStreamSynchronize(c->pfa.stream);
        (void)hipStreamDestroy(c->pfa.stream);
        for (DevBuf *b : {&c->pfa.d_state, &c->pfa.d_rng, &c->pfa.d_off, &c->pfa.d_tcnt, &c->pfa.d_toff})
            if (b->p)
                (void)hipFree(b->p);
        if (c->pfa.h_res)
            (void)hipHostFree(c->pfa.h_res);
        if (c->pfa.e0)
            (void)hipEventDestroy(c->pfa.e0);
        if (c->pfa.e1)
            (void)hipEventDestroy(c->pfa.e1);
    }
    if (c->pf.stream) {
        (void)hipStreamSynchronize(c->pf.stream);
        (void)hipStreamDestroy(c->pf.stream);
        (void)hipHostFree(c->pf.h_count);
        (void)hipEventDestroy(c->pf.e0);
        (void)hipEventDestroy(c->pf.e1);
        for (DevBuf *b : {&c->pf.d_ranges, &c->pf.d_hits, &c->pf.d_counter})
            if (b->p)
                (void)hipFree(b->p);
    }
    for (void *hp : c->host_allocs)
        (void)hipHostFree(hp);
    if (c->up_ring)
        (void)hipHostFree(c->up_ring);
    for (auto &up : c->up_part)
        for (hipEvent_t e : up.ev)
            if (e)
                (void)hipEventDestroy(e);
    for (auto *ln : {&c->l2, &c->l3})
        if (ln->h_lens)
            (void)hipHostFree(ln->h_lens);
    DevBuf *bufs[] = {&c->d_table, &c->d_bloom, &c->d_bloom2, &c->d_sbloom, &c->d_refs, &c->d_ranges, &c->d_hits, &c->d_counter, &c->d_segs, &c->d_slices,
                      &c->d_scratch, &c->d_resv, &c->d_resp, &c->d_dstoff, &c->d_compact, &c->d_jobs, &c->d_counts,
                      &c->d_in, &c->d_pp_cnt, &c->d_pp_off, &c->d_pp_total, &c->d_lag, &c->d_sample, &c->d_zsrc, &c->d_zdst, &c->d_zws,
                      &c->d_zjobs, &c->d_zsize, &c->d_zout, &c->d_zdstoff, &c->d_maybe, &c->d_fjobs,
                      &c->l2.d_segs, &c->l2.d_counter, &c->l2.d_resv, &c->l2.d_resp, &c->l2.d_scratch, &c->l2.d_dstoff,
                      &c->l2.d_compact, &c->l3.d_segs, &c->l3.d_counter, &c->l3.d_resv, &c->l3.d_resp, &c->l3.d_scratch, &c->l3.d_dstoff,
                      &c->l3.d_compact, &c->l2.d_n, &c->l3.d_n, &c->d_esc_jobs, &c->d_flags, &c->d_gmap, &c->d_gmap_stage, &c->d_segwork, &c->d_segtmp};
    if (c->h_segcounts)
        (void)hipHostFree(c->h_segcounts);
    if (c->h_zsizes)
        (void)hipHostFree(c->h_zsizes);
    if (c->h_gmap_stage)
        (void)hipHostFree(c->h_gmap_stage);
    if (c->gmap_ev)
        (void)hipEventDestroy(c->gmap_ev);
    for (PackTemp *t : {&c->pk1, &c->pk_sample, &c->l2.pk, &c->l3.pk})
        for (DevBuf *b : {&t->words, &t->index, &t->esc, &t->cnt})
            if (b->p)
                (void)hipFree(b->p);
    for (DevBuf *b : bufs)
        if (b->p)
            (void)hipFree(b->p);
    for (auto &ch : c->arena)
        (void)hipFree(ch.p);
    for (auto &cbufs : c->chunk_bufs)
        if (cbufs.logs_ahead.valid()) {
            const DevBuf ahead = cbufs.logs_ahead.get();
            if (ahead.p)
                (void)hipFree(ahead.p);
        }
    for (auto &rs : c->ref_store) {
        for (DevBuf *b : {&rs.d_slices, &rs.d_lag, &rs.d_out})
            if (b->p)
                (void)hipFree(b->p);
        if (rs.done)
            (void)hipEventDestroy(rs.done);
    }
    if (c->ref_store_stream) {
        (void)hipStreamSynchronize(c->ref_store_stream);
        (void)hipStreamDestroy(c->ref_store_stream);
    }
    if (c->arena_spare.valid()) {
        const ArenaChunk sp = c->arena_spare.get();
        if (sp.p)
            (void)hipFree(sp.p);
    }
    for (auto &tp : c->tpairs) {
        if (tp.a)
            (void)hipEventDestroy(tp.a);
        if (tp.b)
            (void)hipEventDestroy(tp.b);
    }
    if (c->ev0)
        (void)hipEventDestroy(c->ev0);
    if (c->ev1)
        (void)hipEventDestroy(c->ev1);
    if (c->zev0)
        (void)hipEventDestroy(c->zev0);
    if (c->zev1)
        (void)hipEventDestroy(c->zev1);
    if (c->stream)
        (void)hipStreamDestroy(c->stream);
    if (c->zstream)
        (void)hipStreamDestroy(c->zstream);
    if (c->zstream2)
        (void)hipStreamDestroy(c->zstream2);
    for (hipEvent_t e : {c->zev_a, c->zev_b, c->zev_wait})
        if (e)
            (void)hipEventDestroy(e);
    for (hipEvent_t e : {c->l2.e0, c->l2.e1, c->l2.ready, c->l2.done, c->l3.e0, c->l3.e1, c->l3.ready, c->l3.done})
        if (e)
            (void)hipEventDestroy(e);
    for (hipStream_t s_ : {c->stream2, c->stream3})
        if (s_)
            (void)hipStreamDestroy(s_);
    delete c;
}

const char *agc_hip_last_error(const agc_hip_ctx *c) { return c ? c->err.c_str() : "no context"; }

int agc_hip_sync(agc_hip_ctx *c)
{
    if (!c)
        return AGC_HIP_EINVAL;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return AGC_HIP_OK;
}

int agc_hip_timing_enable(agc_hip_ctx *c, int on)
{
    if (!c)
        return AGC_HIP_EINVAL;
    c->timing = on != 0;
    return AGC_HIP_OK;
}

int agc_hip_timing_reset(agc_hip_ctx *c)
{
    if (!c)
        return AGC_HIP_EINVAL;
    ktimer_read_all(c);
    for (int i = 0; i < AGC_HIP_K_COUNT; ++i) {
        c->ms[i] = 0;
        c->launches[i] = 0;
    }
    return AGC_HIP_OK;
}

int agc_hip_timing_get(agc_hip_ctx *c, int which, double *ms, uint64_t *launches)
{
    if (!c || which < 0 || which >= AGC_HIP_K_COUNT)
        return AGC_HIP_EINVAL;
    ktimer_read_all(c); // (the pairs recorded since the last call: waits for the last of them)
    if (ms)
        *ms = c->ms[which];
    if (launches)
        *launches = c->launches[which];
    return AGC_HIP_OK;
}

int agc_hip_sample_buffer(agc_hip_ctx *c, uint64_t bytes, uint8_t **d_ptr)
{
    if (!c || !d_ptr)
        return AGC_HIP_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    // (the encode of the previous sample may still be reading the buffer on the second lane: see Lane2::done)
    for (auto *ln : {&c->l2, &c->l3})
        if (ln->done_valid) {
            if (bytes + 4096 > c->d_sample.cap)
                HIPCHK(c, hipEventSynchronize(ln->done)); // the buffer is about to be replaced
            else
                HIPCHK(c, hipStreamWaitEvent(c->stream, ln->done, 0));
        }
    for (auto &rs : c->ref_store) // (agc_hip_ref_store_begin_dev does not exist: only packed samples take that path -- kept for symmetry)
        if (rs.pending)
            HIPCHK(c, hipEventSynchronize(rs.done));
    CHK(ensure(c, c->d_sample, bytes + 4096));
    *d_ptr = (uint8_t *)c->d_sample.p;
    return AGC_HIP_OK;
}

int agc_hip_copy_to_device(agc_hip_ctx *c, uint8_t *d_dst, const uint8_t *h_src, uint64_t n)
{
    if (!c || (n && (!d_dst || !h_src)))
        return AGC_HIP_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    if (n) {
        HIPCHK(c, hipMemcpyAsync(d_dst, h_src, n, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return AGC_HIP_OK;
}

int agc_hip_sample_pack(agc_hip_ctx *c, const uint8_t *d_codes, uint64_t n_symbols, agc_hip_packed *out)
{
    if (!c || !out || (n_symbols && !d_codes))
        return AGC_HIP_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    // (the encode of the previous sample may still be reading these buffers on the second lane: see Lane2::done; and the copy of
    // its new references on the reference-store stream)
    for (auto *ln : {&c->l2, &c->l3})
        if (ln->done_valid)
            HIPCHK(c, hipEventSynchronize(ln->done));
    for (auto &rs : c->ref_store)
        if (rs.pending)
            HIPCHK(c, hipEventSynchronize(rs.done));
    PackedSrc src;
    {
        KTimer t(c, AGC_HIP_K_PREPROCESS);
        CHK(pack_bytes(c, d_codes, n_symbols, c->pk_sample, c->stream, src));
    }
    out->d_words = src.words;
    out->d_esc_index = src.esc_index;
    out->d_esc_bytes = src.esc_bytes;
    out->n_symbols = n_symbols;
    return AGC_HIP_OK;
}

// ---------------------------------------------------------------------------
// a1
// ---------------------------------------------------------------------------
int agc_hip_preprocess_dev(agc_hip_ctx *c, const uint8_t *d_raw, uint64_t n_raw, uint8_t *d_codes, uint64_t *h_n_codes)
{
    if (!c || !h_n_codes || (n_raw && (!d_raw || !d_codes)))
        return AGC_HIP_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    *h_n_codes = 0;
    if (!n_raw)
        return AGC_HIP_OK;
    const uint64_t nb64 = (n_raw + PP_TILE - 1) / PP_TILE;
    if (nb64 > 0x7fffffffULL)
        return AGC_HIP_EINVAL;
    const uint32_t nb = (uint32_t)nb64;
    CHK(ensure(c, c->d_pp_cnt, (size_t)nb * 4));
    CHK(ensure(c, c->d_pp_off, (size_t)nb * 8));
    CHK(ensure(c, c->d_pp_total, 8));
    {
        KTimer t(c, AGC_HIP_K_PREPROCESS);
        hipLaunchKernelGGL(pp_count_kernel, dim3(nb), dim3(256), 0, c->stream, d_raw, n_raw, (uint32_t *)c->d_pp_cnt.p);
        hipLaunchKernelGGL(pp_scan_kernel, dim3(1), dim3(1024), 0, c->stream, (const uint32_t *)c->d_pp_cnt.p, nb,
                           (uint64_t *)c->d_pp_off.p, (uint64_t *)c->d_pp_total.p);
        hipLaunchKernelGGL(pp_scatter_kernel, dim3(nb), dim3(256), 0, c->stream, d_raw, n_raw, (const uint64_t *)c->d_pp_off.p, d_codes);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(h_n_codes, c->d_pp_total.p, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return AGC_HIP_OK;
}

int agc_hip_preprocess(agc_hip_ctx *c, const uint8_t *h_raw, uint64_t n_raw, uint8_t *d_codes, uint64_t *h_n_codes)
{
    if (!c || !h_n_codes || (n_raw && (!h_raw || !d_codes)))
        return AGC_HIP_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    *h_n_codes = 0;
    if (!n_raw)
        return AGC_HIP_OK;
    CHK(ensure(c, c->d_in, n_raw + 64));
    HIPCHK(c, hipMemcpyAsync(c->d_in.p, h_raw, n_raw, hipMemcpyHostToDevice, c->stream));
    return agc_hip_preprocess_dev(c, (const uint8_t *)c->d_in.p, n_raw, d_codes, h_n_codes);
}

// ---------------------------------------------------------------------------
// splitters
// ---------------------------------------------------------------------------
static int splitters_upload(agc_hip_ctx *c)
{
    HIPCHK(c, hipSetDevice(c->device));
    uint64_t cap = 1024;
    while (cap < 2 * (uint64_t)c->spl.size())
        cap <<= 1;
    std::vector<uint64_t> tab(cap, ~0ULL);
    std::vector<uint32_t> bloom(BLOOM_WORDS, 0), bloom2(BLOOM2_WORDS, 0);
    for (uint64_t x : c->spl) {
        const uint64_t h = splitter_hash(x);
        uint64_t s = h & (cap - 1);
        while (tab[s] != ~0ULL)
            s = (s + 1) & (cap - 1);
        tab[s] = x;
        uint32_t w, m;
        bloom_slot((uint32_t)(x >> 32), (uint32_t)x, w, m);
        bloom[w] |= m;
        bloom2_slot(h, w, m);
        bloom2[w] |= m;
    }
    CHK(ensure(c, c->d_table, cap * 8));
    CHK(ensure(c, c->d_bloom, BLOOM_WORDS * 4));
    CHK(ensure(c, c->d_bloom2, BLOOM2_WORDS * 4));
    HIPCHK(c, hipMemcpyAsync(c->d_bloom2.p, bloom2.data(), BLOOM2_WORDS * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_table.p, tab.data(), cap * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_bloom.p, bloom.data(), BLOOM_WORDS * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->table_mask = cap - 1;
    return AGC_HIP_OK;
}

int agc_hip_splitters_set(agc_hip_ctx *c, const uint64_t *h_kmers, uint64_t n)
{
    if (!c || (n && !h_kmers))
        return AGC_HIP_EINVAL;
    c->spl.assign(h_kmers, h_kmers + n);
    std::sort(c->spl.begin(), c->spl.end());
    c->spl.erase(std::unique(c->spl.begin(), c->spl.end()), c->spl.end());
    return splitters_upload(c);
}

int agc_hip_splitters_insert(agc_hip_ctx *c, const uint64_t *h_kmers, uint64_t n)
{
    if (!c || (n && !h_kmers))
        return AGC_HIP_EINVAL;
    c->spl.insert(c->spl.end(), h_kmers, h_kmers + n);
    std::sort(c->spl.begin(), c->spl.end());
    c->spl.erase(std::unique(c->spl.begin(), c->spl.end()), c->spl.end());
    return splitters_upload(c);
}

uint64_t agc_hip_splitters_count(const agc_hip_ctx *c) { return c ? c->spl.size() : 0; }

// ---------------------------------------------------------------------------
// scan
// ---------------------------------------------------------------------------
} // extern "C"

// sorts the raw hits and applies the reference's "reset the k-mer after a hit" rule (agc_compressor.cpp:2029): the next hit of the
// same contig must end at least k symbols later
static int deliver_hits(agc_hip_ctx *c, uint32_t n_found, const uint64_t *h_ctg_off, uint32_t n_ctg, uint32_t k, uint64_t cap, uint64_t *h_n_hits,
                        uint32_t *h_hit_ctg, uint64_t *h_hit_pos, uint64_t *h_hit_dir, uint64_t *h_hit_rc, const void *d_hits = nullptr,
                        hipStream_t stream = nullptr)
{
    if (!d_hits) {
        d_hits = c->d_hits.p;
        stream = c->stream;
    }
    std::vector<ScanHit> hits(n_found);
    if (n_found) {
        HIPCHK(c, hipMemcpyAsync(hits.data(), d_hits, (size_t)n_found * sizeof(ScanHit), hipMemcpyDeviceToHost, stream));
        HIPCHK(c, hipStreamSynchronize(stream));
    }
    // by position (positions are unique): LSD radix sort, 11 bits a pass, as many passes as the largest position needs
    if (n_found > 1) {
        uint64_t max_pos = 0;
        for (const ScanHit &h : hits)
            max_pos = std::max<uint64_t>(max_pos, h.pos);
        std::vector<ScanHit> tmp(n_found);
        ScanHit *src = hits.data(), *dst = tmp.data();
        for (int sh = 0; sh < 64 && (max_pos >> sh) != 0; sh += 11) {
            uint32_t cnt[2049] = {0};
            for (uint32_t i = 0; i < n_found; ++i)
                ++cnt[(((uint64_t)src[i].pos >> sh) & 2047u) + 1];
            for (int t = 0; t < 2048; ++t)
                cnt[t + 1] += cnt[t];
            for (uint32_t i = 0; i < n_found; ++i)
                dst[cnt[((uint64_t)src[i].pos >> sh) & 2047u]++] = src[i];
            std::swap(src, dst);
        }
        if (src != hits.data())
            std::memcpy(hits.data(), src, (size_t)n_found * sizeof(ScanHit));
    }

    // accept_hits: after a hit the reference resets the k-mer (agc_compressor.cpp:2029), so the next
    // hit of the same contig must end at least k symbols later.
    uint64_t n_acc = 0;
    uint32_t ci = 0;
    bool have_last = false;
    uint64_t last = 0;
    for (const ScanHit &h : hits) {
        while (ci + 1 < n_ctg && h.pos >= h_ctg_off[ci + 1]) {
            ++ci;
            have_last = false;
        }
        if (have_last && h.pos < last + k)
            continue;
        have_last = true;
        last = h.pos;
        if (n_acc < cap) {
            h_hit_ctg[n_acc] = ci;
            h_hit_pos[n_acc] = h.pos - h_ctg_off[ci];
            h_hit_dir[n_acc] = h.dir;
            h_hit_rc[n_acc] = h.rc;
        }
        ++n_acc;
    }
    *h_n_hits = n_acc;
    return n_acc > cap ? AGC_HIP_ECAP : AGC_HIP_OK;
}


extern "C" {

int agc_hip_scan_contigs_dev(agc_hip_ctx *c, const uint8_t *d_codes, const uint64_t *h_ctg_off, uint32_t n_ctg, uint32_t k,
                             uint64_t cap, uint64_t *h_n_hits, uint32_t *h_hit_ctg, uint64_t *h_hit_pos, uint64_t *h_hit_dir,
                             uint64_t *h_hit_rc)
{
    if (!c || !h_ctg_off || !h_n_hits || k < 2 || k > 32)
        return AGC_HIP_EINVAL;
    if (cap && (!h_hit_ctg || !h_hit_pos || !h_hit_dir || !h_hit_rc))
        return AGC_HIP_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    *h_n_hits = 0;
    if (!c->d_table.p)
        CHK(splitters_upload(c)); // empty set
    const uint64_t total = n_ctg ? h_ctg_off[n_ctg] - h_ctg_off[0] : 0;
    if (!total)
        return AGC_HIP_OK;
    if (!d_codes)
        return AGC_HIP_EINVAL;

    // ranges: contigs cut into pieces of range_len symbols (multiple of one 1 KiB wave step)
    const uint64_t target_waves = 8192ULL * 4;
    uint64_t range_len = (total / target_waves + 1023) / 1024 * 1024;
    range_len = std::min<uint64_t>(std::max<uint64_t>(range_len, 4096), 65536);
    std::vector<ScanRange> ranges;
    for (uint32_t ci = 0; ci < n_ctg; ++ci) {
        const uint64_t b = h_ctg_off[ci], e = h_ctg_off[ci + 1];
        if (e < b)
            return AGC_HIP_EINVAL;
        if (e - b < k)
            continue;
        for (uint64_t p = b; p < e; p += range_len)
            ranges.push_back({b, e, p, std::min(e, p + range_len)});
    }
    if (ranges.empty())
        return AGC_HIP_OK;
    if (ranges.size() > 0x7fffffffULL)
        return AGC_HIP_EINVAL;
    CHK(ensure(c, c->d_ranges, ranges.size() * sizeof(ScanRange)));
    CHK(ensure(c, c->d_counter, 64));
    CHK(upload(c, c->d_ranges.p, ranges.data(), ranges.size() * sizeof(ScanRange), c->stream));

    uint32_t dev_cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(total / 2000 + 4096, c->d_hits.cap / sizeof(ScanHit)), 1u << 30);
    uint32_t n_found = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        CHK(ensure(c, c->d_hits, (size_t)dev_cap * sizeof(ScanHit)));
        HIPCHK(c, hipMemsetAsync(c->d_counter.p, 0, 4, c->stream));
        ScanArgs a;
        a.codes = d_codes;
        a.ranges = (const ScanRange *)c->d_ranges.p;
        a.n_ranges = (uint32_t)ranges.size();
        a.k = k;
        a.table = (const uint64_t *)c->d_table.p;
        a.table_mask = c->table_mask;
        a.bloom = (const uint32_t *)c->d_bloom.p;
        a.bloom2 = (const uint32_t *)c->d_bloom2.p;
        a.hits = (ScanHit *)c->d_hits.p;
        a.n_hits = (uint32_t *)c->d_counter.p;
        a.cap = dev_cap;
        const uint32_t grid = grid_for((uint32_t)ranges.size(), 16, 512);
        {
            KTimer t(c, AGC_HIP_K_SCAN);
            hipLaunchKernelGGL(scan_kernel, dim3(grid), dim3(1024), 0, c->stream, a);
        }
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(&n_found, c->d_counter.p, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (n_found <= dev_cap)
            break;
        dev_cap = n_found;
    }
    return deliver_hits(c, n_found, h_ctg_off, n_ctg, k, cap, h_n_hits, h_hit_ctg, h_hit_pos, h_hit_dir, h_hit_rc);
}

int agc_hip_scan_contigs(agc_hip_ctx *c, const uint8_t *h_codes, const uint64_t *h_ctg_off, uint32_t n_ctg, uint32_t k,
                         uint64_t cap, uint64_t *h_n_hits, uint32_t *h_hit_ctg, uint64_t *h_hit_pos, uint64_t *h_hit_dir,
                         uint64_t *h_hit_rc)
{
    if (!c || !h_ctg_off)
        return AGC_HIP_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    const uint64_t lo = n_ctg ? h_ctg_off[0] : 0, hi = n_ctg ? h_ctg_off[n_ctg] : 0;
    if (hi > lo && !h_codes)
        return AGC_HIP_EINVAL;
    CHK(ensure(c, c->d_in, (hi - lo) + 64));
    if (hi > lo)
        HIPCHK(c, hipMemcpyAsync(c->d_in.p, h_codes + lo, hi - lo, hipMemcpyHostToDevice, c->stream));
    std::vector<uint64_t> off(n_ctg + 1, 0);
    for (uint32_t i = 0; i <= n_ctg && n_ctg; ++i)
        off[i] = h_ctg_off[i] - lo;
    return agc_hip_scan_contigs_dev(c, (const uint8_t *)c->d_in.p, off.data(), n_ctg, k, cap, h_n_hits, h_hit_ctg, h_hit_pos,
                                    h_hit_dir, h_hit_rc);
}


// ---------------------------------------------------------------------------
// 2-bit packed samples
// ---------------------------------------------------------------------------
uint64_t agc_hip_packed_words_bytes(uint64_t n_symbols) { return ((n_symbols + PACK_BLOCK - 1) / PACK_BLOCK) * (PACK_BLOCK / 4) + 64; }
uint64_t agc_hip_packed_index_bytes(uint64_t n_symbols) { return ((n_symbols + PACK_BLOCK - 1) / PACK_BLOCK) * 4 + 64; }

int agc_hip_pack_dev(agc_hip_ctx *c, const uint8_t *d_codes, uint64_t n_symbols, uint32_t *d_words, int32_t *d_esc_index, uint8_t *d_esc_bytes,
                     uint64_t esc_cap_blocks, uint64_t *h_n_esc_blocks)
{
    if (!c || !h_n_esc_blocks || (n_symbols && (!d_codes || !d_words || !d_esc_index)) || (esc_cap_blocks && !d_esc_bytes))
        return AGC_HIP_EINVAL;
    *h_n_esc_blocks = 0;
    if (!n_symbols)
        return AGC_HIP_OK;
    HIPCHK(c, hipSetDevice(c->device));
    CHK(ensure(c, c->d_counter, 64));
    HIPCHK(c, hipMemsetAsync(c->d_counter.p, 0, 4, c->stream));
    const uint64_t n_blocks = (n_symbols + PACK_BLOCK - 1) / PACK_BLOCK;
    {
        KTimer t(c, AGC_HIP_K_PREPROCESS);
        hipLaunchKernelGGL(pack_codes_kernel, dim3((uint32_t)std::min<uint64_t>((n_blocks + 3) / 4, 65536)), dim3(256), 0, c->stream, d_codes, n_symbols,
                           d_words, d_esc_index, d_esc_bytes, (uint32_t *)c->d_counter.p, (uint32_t)std::min<uint64_t>(esc_cap_blocks, 0x7fffffffu));
    }
    HIPCHK(c, hipGetLastError());
    uint32_t cnt = 0;
    HIPCHK(c, hipMemcpyAsync(&cnt, c->d_counter.p, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *h_n_esc_blocks = cnt;
    return cnt > esc_cap_blocks ? AGC_HIP_ECAP : AGC_HIP_OK;
}

int agc_hip_expand_dev(agc_hip_ctx *c, const agc_hip_packed *pk, uint8_t *d_codes)
{
    if (!c || !pk || (pk->n_symbols && (!pk->d_words || !pk->d_esc_index || !d_codes)))
        return AGC_HIP_EINVAL;
    if (!pk->n_symbols)
        return AGC_HIP_OK;
    if ((uintptr_t)d_codes & 15)
        return AGC_HIP_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    const PackedView pv = {pk->d_words, pk->d_esc_index, pk->d_esc_bytes, pk->n_symbols};
    const uint64_t n_blocks = (pk->n_symbols + PACK_BLOCK - 1) / PACK_BLOCK;
    {
        KTimer t(c, AGC_HIP_K_PREPROCESS);
        hipLaunchKernelGGL(expand_codes_kernel, dim3((uint32_t)std::min<uint64_t>((n_blocks + 3) / 4, 65536)), dim3(256), 0, c->stream, pv, d_codes);
    }
    HIPCHK(c, hipGetLastError());
    return AGC_HIP_OK;
}

// raw FASTA bodies (device) -> the packed sample, one pass (pack_kernels.hip), on a stream of its own
int agc_hip_pack_fasta_begin(agc_hip_ctx *c, const uint8_t *d_raw, uint64_t n_raw, const uint64_t *h_raw_begin, const uint64_t *h_raw_end, uint32_t n_ctg,
                             uint32_t *d_words, int32_t *d_esc_index, uint8_t *d_esc_bytes, uint64_t esc_cap_blocks)
{
    if (!c || (n_ctg && (!h_raw_begin || !h_raw_end)) || (n_raw && n_ctg && (!d_raw || !d_words || !d_esc_index)) || (esc_cap_blocks && !d_esc_bytes) ||
        ((uintptr_t)d_raw & 15) || ((uintptr_t)d_words & 15))
        return AGC_HIP_EINVAL;
    uint64_t prev = 0;
    for (uint32_t i = 0; i < n_ctg; ++i) {
        if (h_raw_begin[i] < prev || h_raw_end[i] < h_raw_begin[i] || h_raw_end[i] > n_raw)
            return AGC_HIP_EINVAL;
        prev = h_raw_end[i];
    }
    const uint64_t n_tiles64 = (n_raw + PF_TILE - 1) / PF_TILE;
    if (n_tiles64 > 0x7fffffffULL)
        return AGC_HIP_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    agc_hip_ctx::PackFasta &P = c->pfa;
    if (!P.stream) {
        HIPCHK(c, hipStreamCreateWithFlags(&P.stream, hipStreamNonBlocking));
        HIPCHK(c, hipEventCreate(&P.e0));
        HIPCHK(c, hipEventCreate(&P.e1));
    }
    if (P.pending) { // (a pack nobody collected: dropped)
        HIPCHK(c, hipStreamSynchronize(P.stream));
        P.pending = false;
    }
    P.n_ctg = n_ctg;
    P.n_raw = n_raw;
    P.esc_cap = esc_cap_blocks;
    const size_t res_words = (size_t)n_ctg + 4;
    if (P.h_res_cap < res_words) {
        if (P.h_res)
            HIPCHK(c, hipHostFree(P.h_res));
        P.h_res = nullptr;
        P.h_res_cap = 0;
        HIPCHK(c, hipHostMalloc((void **)&P.h_res, (res_words + 1024) * 8, hipHostMallocDefault));
        P.h_res_cap = res_words + 1024;
    }
    const uint32_t n_tiles = (uint32_t)n_tiles64;
    // the counters: [4, 8) escaped blocks, [8, 16) the total
    CHK(ensure(c, P.d_state, 64, P.stream));
    CHK(ensure(c, P.d_rng, std::max<size_t>(16, (size_t)n_ctg * 16), P.stream));
    CHK(ensure(c, P.d_off, ((size_t)n_ctg + 2) * 8, P.stream));
    HIPCHK(c, hipMemsetAsync(P.d_state.p, 0, 64, P.stream));
    HIPCHK(c, hipMemsetAsync(P.d_off.p, 0xFF, ((size_t)n_ctg + 1) * 8, P.stream));
    if (n_ctg) {
        CHK(upload(c, P.d_rng.p, h_raw_begin, (size_t)n_ctg * 8, P.stream));
        CHK(upload(c, (uint8_t *)P.d_rng.p + (size_t)n_ctg * 8, h_raw_end, (size_t)n_ctg * 8, P.stream));
    }
    P.timed = c->timing;
    if (P.timed)
        (void)hipEventRecord(P.e0, P.stream);
    if (n_tiles && n_ctg) {
        PackFastaArgs a;
        a.raw = d_raw;
        a.n_raw = n_raw;
        a.rng_begin = (const uint64_t *)P.d_rng.p;
        a.rng_end = a.rng_begin + n_ctg;
        a.n_rng = n_ctg;
        a.n_tiles = n_tiles;
        a.esc_count = (uint32_t *)P.d_state.p + 1;
        a.total = (unsigned long long *)P.d_state.p + 1;
        a.words = d_words;
        a.esc_index = d_esc_index;
        a.esc_bytes = d_esc_bytes;
        a.esc_cap = (uint32_t)std::min<uint64_t>(esc_cap_blocks, 0x7fffffffu);
        a.ctg_off = (unsigned long long *)P.d_off.p;
        {
            const uint32_t n_sb = (n_tiles + PF_SCAN_TILES - 1) / PF_SCAN_TILES;
            // d_tcnt: counts per tile, then the scan blocks' totals; d_toff: offsets inside a scan block (u32), then the blocks' offsets (u64)
            const size_t tl_bytes = ((size_t)n_tiles * 4 + 255) & ~(size_t)255;
            CHK(ensure(c, P.d_tcnt, tl_bytes + (size_t)n_sb * 4 + 64, P.stream));
            CHK(ensure(c, P.d_toff, tl_bytes + (size_t)n_sb * 8 + 64, P.stream));
            uint32_t *d_cnt = (uint32_t *)P.d_tcnt.p, *d_tot = (uint32_t *)((uint8_t *)P.d_tcnt.p + tl_bytes);
            uint32_t *d_local = (uint32_t *)P.d_toff.p;
            uint64_t *d_boff = (uint64_t *)((uint8_t *)P.d_toff.p + tl_bytes);
            a.tile_local = d_local;
            a.block_off = d_boff;
            hipLaunchKernelGGL(pack_fasta_count_kernel, dim3(n_tiles), dim3(256), 0, P.stream, a, d_cnt);
            hipLaunchKernelGGL(pack_fasta_scan_kernel, dim3(n_sb), dim3(1024), 0, P.stream, (const uint32_t *)d_cnt, n_tiles, d_local, d_tot);
            hipLaunchKernelGGL(pp_scan_kernel, dim3(1), dim3(1024), 0, P.stream, (const uint32_t *)d_tot, n_sb, d_boff,
                               (uint64_t *)((uint8_t *)P.d_state.p + 32)); // (its total: unused, the last tile writes a.total)
            hipLaunchKernelGGL(pack_fasta_kernel, dim3(n_tiles), dim3(256), 0, P.stream, a);
        }
        HIPCHK(c, hipGetLastError());
    }
    if (P.timed)
        (void)hipEventRecord(P.e1, P.stream);
    HIPCHK(c, hipMemcpyAsync(P.h_res, P.d_off.p, ((size_t)n_ctg + 1) * 8, hipMemcpyDeviceToHost, P.stream));
    HIPCHK(c, hipMemcpyAsync(P.h_res + n_ctg + 1, P.d_state.p, 16, hipMemcpyDeviceToHost, P.stream));
    P.pending = true;
    return AGC_HIP_OK;
}

int agc_hip_pack_fasta_end(agc_hip_ctx *c, uint64_t *h_ctg_off, uint64_t *h_n_esc_blocks)
{
    if (!c || !h_ctg_off || !h_n_esc_blocks)
        return AGC_HIP_EINVAL;
    agc_hip_ctx::PackFasta &P = c->pfa;
    if (!P.pending)
        return AGC_HIP_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(P.stream));
    P.pending = false;
    if (P.timed) {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, P.e0, P.e1);
        c->ms[AGC_HIP_K_PACK] += ms;
        c->launches[AGC_HIP_K_PACK] += 1;
        P.timed = false;
    }
    const uint32_t n = P.n_ctg;
    const uint32_t *cnt = (const uint32_t *)(P.h_res + n + 1); // (unused), escaped blocks; the total in the next 8 bytes
    const uint64_t total = (P.n_raw && n) ? P.h_res[n + 2] : 0;
    for (uint32_t i = 0; i < n; ++i) // (a contig whose first byte no tile holds -- it begins at the end of the buffer -- is empty and begins at the total)
        h_ctg_off[i] = P.h_res[i] == ~0ULL ? total : P.h_res[i];
    h_ctg_off[n] = total;
    *h_n_esc_blocks = (P.n_raw && n) ? cnt[1] : 0;
    return *h_n_esc_blocks > P.esc_cap ? AGC_HIP_ECAP : AGC_HIP_OK;
}

int agc_hip_pack_fasta_dev(agc_hip_ctx *c, const uint8_t *d_raw, uint64_t n_raw, const uint64_t *h_raw_begin, const uint64_t *h_raw_end, uint32_t n_ctg,
                           uint32_t *d_words, int32_t *d_esc_index, uint8_t *d_esc_bytes, uint64_t esc_cap_blocks, uint64_t *h_ctg_off, uint64_t *h_n_esc_blocks)
{
    if (!h_ctg_off || !h_n_esc_blocks)
        return AGC_HIP_EINVAL;
    CHK(agc_hip_pack_fasta_begin(c, d_raw, n_raw, h_raw_begin, h_raw_end, n_ctg, d_words, d_esc_index, d_esc_bytes, esc_cap_blocks));
    return agc_hip_pack_fasta_end(c, h_ctg_off, h_n_esc_blocks);
}

// A window of FASTA bodies in HOST memory -> the packed sample in the context's own buffers (those of agc_hip_sample_pack): the
// bodies go to HBM as they are, back to back, and the pack_fasta kernels make the 2-bit layout from them -- what a host that reads
// FASTA files calls once per window (include/agc_hip.h).
int agc_hip_sample_pack_fasta(agc_hip_ctx *c, uint32_t n_ctg, const uint8_t *const *h_raw, const uint64_t *h_len, agc_hip_packed *out, uint64_t *h_ctg_off)
{
    if (!c || !out || !h_ctg_off || (n_ctg && (!h_raw || !h_len)))
        return AGC_HIP_EINVAL;
    *out = agc_hip_packed{};
    HIPCHK(c, hipSetDevice(c->device));
    if (c->pfa.pending) {
        c->err = "sample_pack_fasta: a conversion queued by agc_hip_pack_fasta_begin is pending";
        return AGC_HIP_EINVAL;
    }
    std::vector<uint64_t> rb(n_ctg), re(n_ctg);
    uint64_t tot = 0;
    for (uint32_t i = 0; i < n_ctg; ++i) {
        if (h_len[i] && !h_raw[i])
            return AGC_HIP_EINVAL;
        rb[i] = tot;
        tot += h_len[i];
        re[i] = tot;
    }
    for (uint32_t i = 0; i <= n_ctg; ++i)
        h_ctg_off[i] = 0;
    if (!tot)
        return AGC_HIP_OK;
    // (the encode of the previous sample may still be reading the packed buffers on a lane: see Lane2::done; and the copy of its new
    // references on the reference-store stream)
    for (auto *ln : {&c->l2, &c->l3})
        if (ln->done_valid)
            HIPCHK(c, hipEventSynchronize(ln->done));
    for (auto &rs : c->ref_store)
        if (rs.pending)
            HIPCHK(c, hipEventSynchronize(rs.done));
    CHK(ensure(c, c->d_in, tot + 64));
    for (uint32_t i = 0; i < n_ctg; ++i)
        if (h_len[i])
            HIPCHK(c, hipMemcpyAsync((uint8_t *)c->d_in.p + rb[i], h_raw[i], h_len[i], hipMemcpyHostToDevice, c->stream));
    // every byte may be a symbol and every block escaped: room for all of them, nothing has to be asked twice
    PackTemp &t = c->pk_sample;
    const uint64_t n_blocks = (tot + PACK_BLOCK - 1) / PACK_BLOCK;
    CHK(ensure(c, t.words, n_blocks * (PACK_BLOCK / 4) + 64));
    CHK(ensure(c, t.index, n_blocks * 4 + 64));
    CHK(ensure(c, t.esc, n_blocks * PACK_BLOCK + 64));
    HIPCHK(c, hipStreamSynchronize(c->stream)); // (the bodies are in HBM, the buffers are where they will stay)
    uint64_t n_esc = 0;
    CHK(agc_hip_pack_fasta_begin(c, (const uint8_t *)c->d_in.p, tot, rb.data(), re.data(), n_ctg, (uint32_t *)t.words.p, (int32_t *)t.index.p, (uint8_t *)t.esc.p,
                                 n_blocks));
    CHK(agc_hip_pack_fasta_end(c, h_ctg_off, &n_esc));
    out->d_words = (const uint32_t *)t.words.p;
    out->d_esc_index = (const int32_t *)t.index.p;
    out->d_esc_bytes = (const uint8_t *)t.esc.p;
    out->n_symbols = h_ctg_off[n_ctg];
    return AGC_HIP_OK;
}

// filter over the last 16 symbols of every splitter and of its reverse complement, for k-mer length k (cached per k)
static int sbloom_upload(agc_hip_ctx *c, uint32_t k)
{
    if (c->sbloom_k == k && c->sbloom_n == c->spl.size() && c->d_sbloom.p)
        return AGC_HIP_OK;
    std::vector<uint32_t> bl(SBLOOM_WORDS, 0);
    const uint32_t lshift = 64 - 2 * k;
    const uint64_t kmask = k == 32 ? ~0ULL : ((1ULL << (2 * k)) - 1ULL);
    auto rev2h = [](uint64_t x) {
        x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
        x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
        return __builtin_bswap64(x);
    };
    for (uint64_t x : c->spl) {
        const uint64_t dir = (x >> lshift) & kmask;                 // right-aligned, last symbol in the low bits
        const uint64_t rc = (rev2h(~dir) >> lshift) & kmask;
        for (uint64_t v : {dir, rc}) {
            uint32_t w, m;
            sbloom_slot((uint32_t)v, w, m);
            bl[w] |= m;
        }
    }
    CHK(ensure(c, c->d_sbloom, SBLOOM_WORDS * 4));
    HIPCHK(c, hipMemcpyAsync(c->d_sbloom.p, bl.data(), SBLOOM_WORDS * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->sbloom_k = k;
    c->sbloom_n = c->spl.size();
    return AGC_HIP_OK;
}

} // extern "C"

// the packed splitter scan of a sample on the context's stream: raw hits (unsorted, before the reset rule) in c->d_hits
static int packed_scan_raw(agc_hip_ctx *c, const agc_hip_packed *pk, const uint64_t *h_ctg_off, uint32_t n_ctg, uint32_t k, uint32_t *n_found_out)
{
    *n_found_out = 0;
    if (!c->d_table.p)
        CHK(splitters_upload(c));
    const uint64_t total = n_ctg ? h_ctg_off[n_ctg] - h_ctg_off[0] : 0;
    if (!total)
        return AGC_HIP_OK;
    if (!pk->d_words || !pk->d_esc_index || h_ctg_off[n_ctg] > pk->n_symbols)
        return AGC_HIP_EINVAL;
    CHK(sbloom_upload(c, k));
    // ranges: pieces of about range_len symbols whose inner boundaries are multiples of PACK_BLOCK in the buffer
    const uint64_t target_waves = 4096ULL * 4;
    uint64_t range_len = (total / target_waves + PACK_BLOCK - 1) / PACK_BLOCK * PACK_BLOCK;
    range_len = std::min<uint64_t>(std::max<uint64_t>(range_len, 8 * PACK_BLOCK), 256 * PACK_BLOCK);
    std::vector<ScanRange> ranges;
    for (uint32_t ci = 0; ci < n_ctg; ++ci) {
        const uint64_t b = h_ctg_off[ci], e = h_ctg_off[ci + 1];
        if (e < b)
            return AGC_HIP_EINVAL;
        if (e - b < k)
            continue;
        for (uint64_t p = b; p < e;) {
            uint64_t q = (p + range_len) & ~(uint64_t)(PACK_BLOCK - 1);
            if (q >= e || e - q < PACK_BLOCK)
                q = e;
            ranges.push_back({b, e, p, q});
            p = q;
        }
    }
    if (ranges.empty())
        return AGC_HIP_OK;
    if (ranges.size() > 0x7fffffffULL)
        return AGC_HIP_EINVAL;
    CHK(ensure(c, c->d_ranges, ranges.size() * sizeof(ScanRange)));
    CHK(ensure(c, c->d_counter, 64));
    CHK(upload(c, c->d_ranges.p, ranges.data(), ranges.size() * sizeof(ScanRange), c->stream));
    if (!c->lds_scan_set) {
        HIPCHK(c, hipFuncSetAttribute((const void *)scan_packed_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SBLOOM_WORDS * 4));
        c->lds_scan_set = true;
    }
    uint32_t dev_cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(total / 2000 + 4096, c->d_hits.cap / sizeof(ScanHit)), 1u << 30);
    uint32_t n_found = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        CHK(ensure(c, c->d_hits, (size_t)dev_cap * sizeof(ScanHit)));
        HIPCHK(c, hipMemsetAsync(c->d_counter.p, 0, 4, c->stream));
        ScanPackedArgs a;
        a.pv = {pk->d_words, pk->d_esc_index, pk->d_esc_bytes, pk->n_symbols};
        a.ranges = (const ScanRange *)c->d_ranges.p;
        a.n_ranges = (uint32_t)ranges.size();
        a.k = k;
        a.table = (const uint64_t *)c->d_table.p;
        a.table_mask = c->table_mask;
        a.sbloom = (const uint32_t *)c->d_sbloom.p;
        a.bloom2 = (const uint32_t *)c->d_bloom2.p;
        a.hits = (ScanHit *)c->d_hits.p;
        a.n_hits = (uint32_t *)c->d_counter.p;
        a.cap = dev_cap;
        const uint32_t grid = grid_for((uint32_t)ranges.size(), 16, 256); // one 1024-thread block per CU (128 KiB of LDS each)
        {
            KTimer t(c, AGC_HIP_K_SCAN);
            hipLaunchKernelGGL(scan_packed_kernel, dim3(grid), dim3(1024), SBLOOM_WORDS * 4, c->stream, a);
        }
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(&n_found, c->d_counter.p, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (n_found <= dev_cap)
            break;
        dev_cap = n_found;
    }
    *n_found_out = n_found;
    return AGC_HIP_OK;
}

extern "C" {

int agc_hip_scan_packed_dev(agc_hip_ctx *c, const agc_hip_packed *pk, const uint64_t *h_ctg_off, uint32_t n_ctg, uint32_t k, uint64_t cap,
                            uint64_t *h_n_hits, uint32_t *h_hit_ctg, uint64_t *h_hit_pos, uint64_t *h_hit_dir, uint64_t *h_hit_rc)
{
    if (!c || !pk || !h_ctg_off || !h_n_hits || k < 16 || k > 32)
        return AGC_HIP_EINVAL; // (k < 16: the last-16-symbols filter does not apply; expand and use agc_hip_scan_contigs_dev)
    if (cap && (!h_hit_ctg || !h_hit_pos || !h_hit_dir || !h_hit_rc))
        return AGC_HIP_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    *h_n_hits = 0;
    uint32_t n_found = 0;
    CHK(packed_scan_raw(c, pk, h_ctg_off, n_ctg, k, &n_found));
    if (!n_found)
        return AGC_HIP_OK;
    return deliver_hits(c, n_found, h_ctg_off, n_ctg, k, cap, h_n_hits, h_hit_ctg, h_hit_pos, h_hit_dir, h_hit_rc);
}

// The next sample ahead of its turn (include/agc_hip.h): its packed scan queued on the prefetch stream, nothing waited for.
int agc_hip_prefetch_packed_dev(agc_hip_ctx *c, const agc_hip_packed *pk, const uint64_t *h_ctg_off, uint32_t n_ctg, uint32_t k)
{
    if (!c || !pk || !h_ctg_off || !n_ctg || k < 16 || k > 32 || !pk->d_words || !pk->d_esc_index || !pk->n_symbols ||
        h_ctg_off[n_ctg] > pk->n_symbols)
        return AGC_HIP_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    agc_hip_ctx::Prefetch &pf = c->pf;
    if (!pf.stream) {
        HIPCHK(c, hipStreamCreateWithFlags(&pf.stream, hipStreamNonBlocking));
        HIPCHK(c, hipHostMalloc((void **)&pf.h_count, 64, hipHostMallocDefault));
        HIPCHK(c, hipEventCreate(&pf.e0));
        HIPCHK(c, hipEventCreate(&pf.e1));
    }
    HIPCHK(c, hipStreamSynchronize(pf.stream)); // (a prefetch nobody asked for again: its kernels must be done before its buffers go)
    pf.valid = false;
    if (!c->d_table.p)
        CHK(splitters_upload(c));
    CHK(sbloom_upload(c, k));
    const PackedView pv = {pk->d_words, pk->d_esc_index, pk->d_esc_bytes, pk->n_symbols};
    pf.timed = c->timing;
    if (pf.timed)
        HIPCHK(c, hipEventRecord(pf.e0, pf.stream));
    // the scan: as agc_hip_scan_packed_dev, own scratch, nothing waited for
    const uint64_t total = h_ctg_off[n_ctg] - h_ctg_off[0];
    const uint64_t target_waves = 4096ULL * 4;
    uint64_t range_len = (total / target_waves + PACK_BLOCK - 1) / PACK_BLOCK * PACK_BLOCK;
    range_len = std::min<uint64_t>(std::max<uint64_t>(range_len, 8 * PACK_BLOCK), 256 * PACK_BLOCK);
    std::vector<ScanRange> ranges;
    for (uint32_t ci = 0; ci < n_ctg; ++ci) {
        const uint64_t b = h_ctg_off[ci], e = h_ctg_off[ci + 1];
        if (e < b)
            return AGC_HIP_EINVAL;
        if (e - b < k)
            continue;
        for (uint64_t p = b; p < e;) {
            uint64_t q = (p + range_len) & ~(uint64_t)(PACK_BLOCK - 1);
            if (q >= e || e - q < PACK_BLOCK)
                q = e;
            ranges.push_back({b, e, p, q});
            p = q;
        }
    }
    pf.scanned = !ranges.empty() && ranges.size() <= 0x7fffffffULL;
    *pf.h_count = 0;
    if (pf.scanned) {
        CHK(ensure(c, pf.d_ranges, ranges.size() * sizeof(ScanRange), pf.stream));
        CHK(ensure(c, pf.d_counter, 64, pf.stream));
        pf.dev_cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(total / 1000 + 4096, pf.d_hits.cap / sizeof(ScanHit)), 1u << 30);
        CHK(ensure(c, pf.d_hits, (size_t)pf.dev_cap * sizeof(ScanHit), pf.stream));
        // (the ranges go through a pageable vector: the copy is staged by the runtime before the call returns)
        CHK(upload(c, pf.d_ranges.p, ranges.data(), ranges.size() * sizeof(ScanRange), pf.stream));
        HIPCHK(c, hipMemsetAsync(pf.d_counter.p, 0, 4, pf.stream));
        if (!c->lds_scan_set) {
            HIPCHK(c, hipFuncSetAttribute((const void *)scan_packed_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SBLOOM_WORDS * 4));
            c->lds_scan_set = true;
        }
        ScanPackedArgs a;
        a.pv = pv;
        a.ranges = (const ScanRange *)pf.d_ranges.p;
        a.n_ranges = (uint32_t)ranges.size();
        a.k = k;
        a.table = (const uint64_t *)c->d_table.p;
        a.table_mask = c->table_mask;
        a.sbloom = (const uint32_t *)c->d_sbloom.p;
        a.bloom2 = (const uint32_t *)c->d_bloom2.p;
        a.hits = (ScanHit *)pf.d_hits.p;
        a.n_hits = (uint32_t *)pf.d_counter.p;
        a.cap = pf.dev_cap;
        const uint32_t grid = grid_for((uint32_t)ranges.size(), 16, 256);
        hipLaunchKernelGGL(scan_packed_kernel, dim3(grid), dim3(1024), SBLOOM_WORDS * 4, pf.stream, a);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(pf.h_count, pf.d_counter.p, 4, hipMemcpyDeviceToHost, pf.stream));
    }
    if (pf.timed)
        HIPCHK(c, hipEventRecord(pf.e1, pf.stream));
    pf.words = pk->d_words;
    pf.n_symbols = pk->n_symbols;
    pf.k = k;
    pf.n_ctg = n_ctg;
    pf.first_off = h_ctg_off[0];
    pf.last_off = h_ctg_off[n_ctg];
    pf.valid = true;
    return AGC_HIP_OK;
}

int agc_hip_scan_prefetched(agc_hip_ctx *c, const agc_hip_packed *pk, const uint64_t *h_ctg_off, uint32_t n_ctg, uint32_t k, uint64_t cap,
                            uint64_t *h_n_hits, uint32_t *h_hit_ctg, uint64_t *h_hit_pos, uint64_t *h_hit_dir, uint64_t *h_hit_rc)
{
    if (!c || !pk || !h_ctg_off || !h_n_hits)
        return AGC_HIP_EINVAL;
    agc_hip_ctx::Prefetch &pf = c->pf;
    if (!pf.valid || pf.words != pk->d_words || pf.n_symbols != pk->n_symbols || pf.k != k || pf.n_ctg != n_ctg || pf.first_off != h_ctg_off[0] ||
        pf.last_off != h_ctg_off[n_ctg])
        return AGC_HIP_EINVAL; // nothing, or something else, was prefetched
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(pf.stream));
    if (pf.timed) { // (the prefetched scan: the scan row)
        float ms = 0;
        (void)hipEventElapsedTime(&ms, pf.e0, pf.e1);
        c->ms[AGC_HIP_K_SCAN] += ms;
        c->launches[AGC_HIP_K_SCAN] += 1;
        pf.timed = false;
    }
    *h_n_hits = 0;
    if (!pf.scanned)
        return AGC_HIP_OK;
    const uint32_t n_found = *pf.h_count;
    if (n_found > pf.dev_cap) // (more hits than the list held: the scan again, the ordinary way)
        return agc_hip_scan_packed_dev(c, pk, h_ctg_off, n_ctg, k, cap, h_n_hits, h_hit_ctg, h_hit_pos, h_hit_dir, h_hit_rc);
    return deliver_hits(c, n_found, h_ctg_off, n_ctg, k, cap, h_n_hits, h_hit_ctg, h_hit_pos, h_hit_dir, h_hit_rc, pf.d_hits.p, pf.stream);
}

// ---------------------------------------------------------------------------
// references
// ---------------------------------------------------------------------------
} // extern "C"

// new references out of a packed buffer: stored form (2-bit words from bit 0, escape blocks where needed), index, key filter
static int ref_register_impl(agc_hip_ctx *c, uint32_t n_refs, const uint32_t *h_gid, const PackedSrc &src, const uint64_t *h_off,
                             const uint32_t *h_len, const uint8_t *h_rc, uint32_t min_match_len)
{
    if (min_match_len < HASHING_STEP + 4 || min_match_len > 32)
        return AGC_HIP_EINVAL; // key_len = mml-3 must fit 2 bits x key_len <= 58 and the wave's 64 lanes
    if (!n_refs)
        return AGC_HIP_OK;
    const uint32_t key_len = min_match_len - HASHING_STEP + 1;
    uint32_t max_gid = 0;
    for (uint32_t i = 0; i < n_refs; ++i) {
        max_gid = std::max(max_gid, h_gid[i]);
        if (h_gid[i] < c->refs.size() && c->refs[h_gid[i]].valid) {
            c->err = "group " + std::to_string(h_gid[i]) + " already has a reference";
            return AGC_HIP_EINVAL;
        }
    }
    {
        std::vector<uint32_t> g(h_gid, h_gid + n_refs);
        std::sort(g.begin(), g.end());
        if (std::adjacent_find(g.begin(), g.end()) != g.end())
            return AGC_HIP_EINVAL;
    }

    // 1. stored form: ref_size symbols in 2-bit words + REF_TAIL_WORDS words of slack, 16-byte aligned
    std::vector<IdxBuild> jobs(n_refs);
    size_t ref_bytes = 0;
    std::vector<size_t> roff(n_refs);
    for (uint32_t i = 0; i < n_refs; ++i) {
        roff[i] = ref_bytes;
        ref_bytes += ((((size_t)h_len[i] + 15) / 16 + REF_TAIL_WORDS) * 4 + 15) & ~(size_t)15;
    }
    uint8_t *rbase = nullptr;
    CHK(arena_alloc(c, ref_bytes, &rbase));
    for (uint32_t i = 0; i < n_refs; ++i) {
        jobs[i].src = {src.words, src.esc_index, src.esc_bytes, h_off[i], h_len[i], h_rc ? (uint32_t)(h_rc[i] != 0) : 0u};
        jobs[i].words = (uint32_t *)(rbase + roff[i]);
        jobs[i].esc_index = nullptr;
        jobs[i].esc_bytes = nullptr;
        jobs[i].table = nullptr;
        jobs[i].bloom = nullptr;
        jobs[i].ref_size = h_len[i];
        jobs[i].key_len = key_len;
        jobs[i].ht_mask = 0;
        jobs[i].is_short = (h_len[i] / HASHING_STEP) < 65535u; // lz_diff.cpp:146
    }
    CHK(ensure(c, c->d_jobs, (size_t)n_refs * sizeof(IdxBuild)));
    CHK(ensure(c, c->d_counts, (size_t)n_refs * 4));
    CHK(ensure(c, c->d_flags, (size_t)n_refs * 4));
    CHK(upload(c, c->d_jobs.p, jobs.data(), (size_t)n_refs * sizeof(IdxBuild), c->stream));
    // few references per batch (steady state): spread each over several blocks to fill the chip
    const uint32_t split = n_refs >= 2048 ? 1u : std::min<uint32_t>(16u, 2048u / n_refs);
    HIPCHK(c, hipMemsetAsync(c->d_flags.p, 0, (size_t)n_refs * 4, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_counts.p, 0, (size_t)n_refs * 4, c->stream));
    {
        KTimer t(c, AGC_HIP_K_REFSTORE);
        hipLaunchKernelGGL(ref_pack_kernel, dim3(n_refs * split), dim3(256), 0, c->stream, (const IdxBuild *)c->d_jobs.p, (uint32_t *)c->d_flags.p, split);
    }
    // 2. count keys -> table sizes (prepare_index, lz_diff.cpp:81-125: double division by 0.7,
    //    round down to a power of two, double it, at least 8)
    std::vector<uint32_t> counts(n_refs), flags(n_refs);
    {
        KTimer t(c, AGC_HIP_K_INDEX);
        hipLaunchKernelGGL(idx_count_kernel, dim3(n_refs * split), dim3(256), 0, c->stream, (const IdxBuild *)c->d_jobs.p,
                           (uint32_t *)c->d_counts.p, split);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(counts.data(), c->d_counts.p, (size_t)n_refs * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(flags.data(), c->d_flags.p, (size_t)n_refs * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    size_t tab_bytes = 0;
    std::vector<size_t> toff(n_refs);
    for (uint32_t i = 0; i < n_refs; ++i) {
        uint64_t hs = (uint64_t)((double)counts[i] / 0.7);
        while (hs & (hs - 1))
            hs &= hs - 1;
        hs <<= 1;
        if (hs < 8)
            hs = 8;
        jobs[i].ht_mask = (uint32_t)(hs - 1);
        toff[i] = tab_bytes;
        tab_bytes += (size_t)hs * (jobs[i].is_short ? 4 : 8);
        tab_bytes = (tab_bytes + 255) & ~(size_t)255;
    }
    uint8_t *tbase = nullptr;
    CHK(arena_alloc(c, tab_bytes, &tbase));
    HIPCHK(c, hipMemsetAsync(tbase, 0xFF, tab_bytes, c->stream));
    // key filters (two per reference, sized by its length: dev_common.h; zeroed; the insert kernel sets the bits)
    uint8_t *bbase = nullptr;
    std::vector<size_t> boff(n_refs);
    size_t bloom_bytes = 0;
    for (uint32_t i = 0; i < n_refs; ++i) {
        boff[i] = bloom_bytes;
        bloom_bytes += (size_t)2 * key_bloom_half_words(key_bloom_shift(jobs[i].ref_size)) * 8;
    }
    CHK(arena_alloc(c, bloom_bytes, &bbase));
    HIPCHK(c, hipMemsetAsync(bbase, 0, bloom_bytes, c->stream));
    // references that hold a symbol outside ACGT: escape index + one byte per symbol for the blocks that need it
    std::vector<EscJob> ejobs;
    for (uint32_t i = 0; i < n_refs; ++i) {
        jobs[i].table = tbase + toff[i];
        jobs[i].bloom = (unsigned long long *)(bbase + boff[i]);
        if (flags[i]) {
            const uint32_t nb = (h_len[i] + PACK_BLOCK - 1) / PACK_BLOCK;
            uint8_t *eb = nullptr;
            CHK(arena_alloc(c, (size_t)nb * 4 + 64 + (size_t)nb * PACK_BLOCK, &eb));
            jobs[i].esc_index = (int32_t *)eb;
            jobs[i].esc_bytes = eb + (((size_t)nb * 4 + 64 + 15) & ~(size_t)15);
            for (uint32_t b = 0; b < nb; ++b)
                ejobs.push_back({jobs[i].src, jobs[i].esc_index, jobs[i].esc_bytes, b, 0u});
        }
    }
    if (!ejobs.empty()) {
        CHK(ensure(c, c->d_esc_jobs, ejobs.size() * sizeof(EscJob)));
        CHK(upload(c, c->d_esc_jobs.p, ejobs.data(), ejobs.size() * sizeof(EscJob), c->stream));
        KTimer t(c, AGC_HIP_K_REFSTORE);
        hipLaunchKernelGGL(ref_esc_kernel, dim3((uint32_t)ejobs.size()), dim3(256), 0, c->stream, (const EscJob *)c->d_esc_jobs.p);
    }
    CHK(upload(c, c->d_jobs.p, jobs.data(), (size_t)n_refs * sizeof(IdxBuild), c->stream));
    {
        KTimer t(c, AGC_HIP_K_INDEX);
        hipLaunchKernelGGL(idx_insert_kernel, dim3(n_refs * split), dim3(256), 0, c->stream, (const IdxBuild *)c->d_jobs.p, split);
    }
    HIPCHK(c, hipGetLastError());
    // (whatever parses against these tables is ordered behind the insert -- the steps' stream itself, or a lane through the `ready`
    // event its begin records there -- so this wait is not needed for them; without it the step measured the same, twice: round 6)
    HIPCHK(c, hipStreamSynchronize(c->stream));

    if (c->refs.size() <= max_gid)
        c->refs.resize((size_t)max_gid + 1, RefDesc{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 0});
    for (uint32_t i = 0; i < n_refs; ++i) {
        RefDesc &r = c->refs[h_gid[i]];
        r.words = jobs[i].words;
        r.esc_index = jobs[i].esc_index;
        r.esc_bytes = jobs[i].esc_bytes;
        r.table = jobs[i].table;
        r.bloom = jobs[i].bloom;
        r.ref_size = h_len[i];
        r.ht_mask = jobs[i].ht_mask;
        r.key_len = key_len;
        r.min_match_len = min_match_len;
        r.is_short = jobs[i].is_short;
        r.valid = 1;
    }
    {
        uint32_t min_gid = max_gid;
        for (uint32_t i = 0; i < n_refs; ++i)
            min_gid = std::min(min_gid, h_gid[i]);
        if (!c->refs_dirty || c->refs_dirty_hi <= c->refs_dirty_lo) {
            c->refs_dirty_lo = min_gid;
            c->refs_dirty_hi = (size_t)max_gid + 1;
        } else {
            c->refs_dirty_lo = std::min<size_t>(c->refs_dirty_lo, min_gid);
            c->refs_dirty_hi = std::max<size_t>(c->refs_dirty_hi, (size_t)max_gid + 1);
        }
        // (descriptors the resize above added between the old end and min_gid are invalid ones: they go over too)
        c->refs_dirty_lo = std::min(c->refs_dirty_lo, c->refs_on_dev);
    }
    c->refs_dirty = true;
    return AGC_HIP_OK;
}

extern "C" {

int agc_hip_ref_register_batch_packed(agc_hip_ctx *c, uint32_t n_refs, const uint32_t *h_gid, const agc_hip_packed *pk, const uint64_t *h_off,
                                      const uint32_t *h_len, const uint8_t *h_rc, uint32_t min_match_len)
{
    if (!c || (n_refs && (!h_gid || !h_off || !h_len || !pk || !pk->d_words)))
        return AGC_HIP_EINVAL;
    if (!n_refs)
        return AGC_HIP_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const PackedSrc src = src_of(pk);
    if (!slices_inside(src, n_refs, h_off, h_len))
        return AGC_HIP_EINVAL;
    return ref_register_impl(c, n_refs, h_gid, src, h_off, h_len, h_rc, min_match_len);
}

int agc_hip_ref_register_batch_dev(agc_hip_ctx *c, uint32_t n_refs, const uint32_t *h_gid, const uint8_t *d_base,
                                   const uint64_t *h_off, const uint32_t *h_len, const uint8_t *h_rc, uint32_t min_match_len)
{
    if (!c || (n_refs && (!h_gid || !h_off || !h_len || !d_base)))
        return AGC_HIP_EINVAL;
    if (!n_refs)
        return AGC_HIP_OK;
    HIPCHK(c, hipSetDevice(c->device));
    PackedSrc src;
    std::vector<uint64_t> off2;
    CHK(pack_range(c, d_base, n_refs, h_off, h_len, c->pk1, c->stream, src, off2));
    return ref_register_impl(c, n_refs, h_gid, src, off2.data(), h_len, h_rc, min_match_len);
}

int agc_hip_ref_register(agc_hip_ctx *c, uint32_t gid, const uint8_t *h_ref, uint32_t n, uint32_t min_match_len)
{
    if (!c || (n && !h_ref))
        return AGC_HIP_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    CHK(ensure(c, c->d_in, (size_t)n + 64));
    if (n)
        HIPCHK(c, hipMemcpyAsync(c->d_in.p, h_ref, n, hipMemcpyHostToDevice, c->stream));
    const uint64_t off = 0;
    const uint8_t rc = 0;
    return agc_hip_ref_register_batch_dev(c, 1, &gid, (const uint8_t *)c->d_in.p, &off, &n, &rc, min_match_len);
}

int agc_hip_ref_get(agc_hip_ctx *c, uint32_t gid, uint8_t *h_ref, uint32_t cap, uint32_t *h_n)
{
    if (!c || !h_n)
        return AGC_HIP_EINVAL;
    if (gid >= c->refs.size() || !c->refs[gid].valid)
        return AGC_HIP_ENOREF;
    HIPCHK(c, hipSetDevice(c->device));
    const RefDesc &r = c->refs[gid];
    *h_n = r.ref_size;
    if (cap < r.ref_size)
        return AGC_HIP_ECAP;
    if (r.ref_size) { // the stored 2-bit form back as one byte per symbol
        CHK(ensure(c, c->d_compact, (size_t)r.ref_size + 64));
        const ViewJob vj = {{r.words, r.esc_index, r.esc_bytes, 0, r.ref_size, 0}, c->d_compact.p};
        CHK(ensure(c, c->d_slices, sizeof(ViewJob)));
        HIPCHK(c, hipMemcpyAsync(c->d_slices.p, &vj, sizeof(ViewJob), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(slice_expand_kernel, dim3(1), dim3(256), 0, c->stream, (const ViewJob *)c->d_slices.p, 1u);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(h_ref, c->d_compact.p, r.ref_size, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return AGC_HIP_OK;
}

int agc_hip_ref_index_get(agc_hip_ctx *c, uint32_t gid, uint32_t *h_table, uint64_t cap, uint64_t *h_ht_size, int *h_is16)
{
    if (!c || !h_ht_size || !h_is16)
        return AGC_HIP_EINVAL;
    if (gid >= c->refs.size() || !c->refs[gid].valid)
        return AGC_HIP_ENOREF;
    HIPCHK(c, hipSetDevice(c->device));
    const RefDesc &r = c->refs[gid];
    const uint64_t hs = (uint64_t)r.ht_mask + 1;
    *h_ht_size = hs;
    *h_is16 = r.is_short;
    if (cap < hs)
        return AGC_HIP_ECAP;
    HIPCHK(c, hipStreamSynchronize(c->stream)); // (the insert of the registration may still be running)
    if (r.is_short) {
        std::vector<uint32_t> t(hs);
        HIPCHK(c, hipMemcpy(t.data(), r.table, hs * 4, hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < hs; ++i)
            h_table[i] = t[i] == 0xFFFFFFFFu ? 0xFFFFu : (t[i] >> 16);
    } else {
        std::vector<uint64_t> t(hs);
        HIPCHK(c, hipMemcpy(t.data(), r.table, hs * 8, hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < hs; ++i)
            h_table[i] = t[i] == ~0ULL ? 0xFFFFFFFFu : (uint32_t)(t[i] >> 32);
    }
    return AGC_HIP_OK;
}

// ---------------------------------------------------------------------------
// parse batches
// ---------------------------------------------------------------------------
} // extern "C"

namespace {

struct Batch {
    std::vector<SegDesc> segs; // processing order (longest first); SegDesc.pad = original index
    uint64_t out_total = 0;    // scratch units (bytes or u32s)
};

// Builds descriptors: every text is a view into the packed buffer (either orientation -- nothing is copied or staged).
int prepare_batch(agc_hip_ctx *c, int mode, uint32_t n, const uint32_t *h_gid, const PackedSrc &src, const uint64_t *h_off,
                  const uint32_t *h_len, const uint8_t *h_rc, const uint8_t *h_prefix, Batch &b, int lane = 0)
{
    // (lane 2: encode only -- no filter bitmaps; its own scratch and stream)
    // (lane: 0 = the first stream and its buffers; 1, 2 = an encode lane of its own, agc_hip_ctx::lane)
    DevBuf &L_segs = lane ? c->lane(lane).d_segs : c->d_segs, &L_counter = lane ? c->lane(lane).d_counter : c->d_counter,
           &L_resv = lane ? c->lane(lane).d_resv : c->d_resv, &L_resp = lane ? c->lane(lane).d_resp : c->d_resp;
    const hipStream_t L_stream = lane ? c->lane(lane).s : c->stream;
    static const bool laps = getenv("AGC_HIP_LAPS") != nullptr;
    auto tnow = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double lt = laps ? tnow() : 0;
    auto LAP = [&](const char *what) {
        if (laps && n > 10000) {
            const double t = tnow();
            fprintf(stderr, "    prepare_batch lap %s %.3f ms\n", what, t - lt);
            lt = t;
        }
    };
    for (uint32_t i = 0; i < n; ++i)
        if (h_gid[i] >= c->refs.size() || !c->refs[h_gid[i]].valid) {
            c->err = "group " + std::to_string(h_gid[i]) + " has no registered reference";
            return AGC_HIP_ENOREF;
        }
    if (!slices_inside(src, n, h_off, h_len)) {
        c->err = "a sequence runs past the end of its buffer";
        return AGC_HIP_EINVAL;
    }
    LAP("valid");
    CHK(upload_refs(c));
    // longest first, stable in the index: LSD radix sort on ~len (3 passes of 11 bits; a comparison sort of the 50 k segments
    // of a human sample costs milliseconds of host time per call)
    std::vector<uint32_t> order(n);
    {
        std::vector<uint32_t> tmp(n);
        for (uint32_t i = 0; i < n; ++i)
            order[i] = i;
        uint32_t *src = order.data(), *dst = tmp.data();
        for (int pass = 0; pass < 3; ++pass) {
            const int sh = 11 * pass;
            uint32_t cnt[2049] = {0};
            for (uint32_t i = 0; i < n; ++i)
                ++cnt[((~h_len[src[i]] >> sh) & 2047u) + 1];
            for (int t = 0; t < 2048; ++t)
                cnt[t + 1] += cnt[t];
            for (uint32_t i = 0; i < n; ++i)
                dst[cnt[(~h_len[src[i]] >> sh) & 2047u]++] = src[i];
            std::swap(src, dst);
        }
        if (src != order.data())
            std::memcpy(order.data(), src, (size_t)n * 4);
    }
    LAP("sort");
    if (laps && mode == MODE_COSTVEC && n) {
        uint64_t tot = 0;
        for (uint32_t i = 0; i < n; ++i)
            tot += h_len[i];
        fprintf(stderr, "    cost-vector batch: %u parses, %.1f Mb of text, longest %u %u %u %u, median %u\n", n, tot / 1e6, h_len[order[0]], h_len[order[n > 1 ? 1 : 0]],
                h_len[order[n > 2 ? 2 : 0]], h_len[order[n > 3 ? 3 : 0]], h_len[order[n / 2]]);
    }
    std::vector<uint64_t> ooff(n);
    uint64_t tot = 0;
    for (uint32_t i = 0; i < n; ++i) {
        ooff[i] = tot;
        if (mode == MODE_ENCODE)
            tot += (((uint64_t)h_len[i] + 5ULL * h_len[i] / 16 + 64) + 15) & ~15ULL; // a >=16-symbol match costs <= 21 bytes
        else if (mode == MODE_COSTVEC)
            tot += h_len[i];
    }
    b.out_total = tot;
    b.segs.resize(n);
    // estimate / cost vector: one "may match" bit per text position (key_filter_kernel) lets the parse skip literal runs
    std::vector<FilterJob> fjobs;
    std::vector<uint64_t> moff(n, 0);
    if (mode != MODE_ENCODE) {
        uint64_t words = 0;
        for (uint32_t i = 0; i < n; ++i) {
            moff[i] = words;
            words += ((uint64_t)h_len[i] + 63) / 64 + 1;
        }
        CHK(ensure(c, c->d_maybe, words * 8 + 64));
    }
    for (uint32_t p = 0; p < n; ++p) {
        const uint32_t i = order[p];
        SegDesc &s = b.segs[p];
        s.maybe = nullptr;
        s.text = {src.words, src.esc_index, src.esc_bytes, h_off[i], h_len[i], (h_rc && h_rc[i]) ? 1u : 0u};
        s.out_off = ooff[i];
        s.ref_slot = h_gid[i];
        s.flags = (h_prefix && h_prefix[i]) ? 1u : 0u;
        s.idx = i;
        s.pad = 0;
        const RefDesc &rd = c->refs[h_gid[i]];
        if (mode != MODE_ENCODE && rd.bloom && h_len[i] > 4 * WAVE) { // (short texts: not worth a block)
            s.maybe = (const unsigned long long *)c->d_maybe.p + moff[i];
            for (uint32_t ch = 0; ch < h_len[i]; ch += FILTER_CHUNK)
                fjobs.push_back({s.text, rd.bloom, (unsigned long long *)c->d_maybe.p + moff[i], rd.key_len, ch, key_bloom_shift(rd.ref_size), 0u});
        }
    }
    if (!fjobs.empty()) {
        CHK(ensure(c, c->d_fjobs, fjobs.size() * sizeof(FilterJob)));
        CHK(upload(c, c->d_fjobs.p, fjobs.data(), fjobs.size() * sizeof(FilterJob), L_stream));
        {
            KTimer t(c, AGC_HIP_K_FILTER);
            hipLaunchKernelGGL(key_filter_kernel, dim3((uint32_t)fjobs.size()), dim3(FILTER_THREADS), 0, L_stream, (const FilterJob *)c->d_fjobs.p);
        }
        HIPCHK(c, hipGetLastError()); // (fjobs is a local: upload() took its copy)
    }
    LAP("descriptors");
    CHK(ensure(c, L_segs, (size_t)n * sizeof(SegDesc), L_stream));
    CHK(upload(c, L_segs.p, b.segs.data(), (size_t)n * sizeof(SegDesc), L_stream));
    CHK(ensure(c, L_counter, 64, L_stream));
    HIPCHK(c, hipMemsetAsync(L_counter.p, 0, 4, L_stream));
    CHK(ensure(c, L_resv, (size_t)n * 4, L_stream));
    CHK(ensure(c, L_resp, (size_t)n * 4, L_stream));
    LAP("upload");
    return AGC_HIP_OK;
}

// A launch with few, long texts is a handful of dependent chains on an empty GPU: such batches are parsed in chunks (lz_kernels.hip:
// ChunkCtl) -- a wavefront per 4096-symbol chunk, then a wavefront per text that joins them.  AGC_HIP_LZ_CHUNK=<symbols> forces a chunk
// length for every launch the host has descriptors of (the tests run the whole LZ suite that way), 0 switches the path off.
template <int MODE>
int launch_parse(agc_hip_ctx *c, uint32_t n, uint8_t *out_bytes, uint32_t *out_u32, int lane = 0, const uint32_t *n_dev = nullptr, const Batch *b = nullptr)
{
    const uint32_t grid = (n + 3) / 4; // one wave per segment, 4 waves per block
    const hipStream_t st = lane ? c->lane(lane).s : c->stream;
    const RefDesc *d_refs = (const RefDesc *)c->d_refs.p;
    const SegDesc *d_segs = (const SegDesc *)(lane ? c->lane(lane).d_segs.p : c->d_segs.p);
    uint32_t *d_resv = (uint32_t *)(lane ? c->lane(lane).d_resv.p : c->d_resv.p), *d_resp = (uint32_t *)(lane ? c->lane(lane).d_resp.p : c->d_resp.p);
    static const int forced = getenv("AGC_HIP_LZ_CHUNK") ? atoi(getenv("AGC_HIP_LZ_CHUNK")) : -1;
    uint32_t chunk_len = 0;
    if (b && !n_dev && n && forced != 0 && b->segs.size() == n) {
        if (forced > 0)
            chunk_len = (uint32_t)std::max(64, forced);
        else if (b->segs[0].text.len >= 16384) { // (longest first)
            // worth it when the longest text's chain outlasts the launch's work spread over the chip (~6 k wavefronts in flight):
            // the diverged collections' few long texts, not the 3 000 cost-vector parses of a human sample
            uint64_t total = 0;
            for (const SegDesc &sd : b->segs)
                total += sd.text.len;
            if ((uint64_t)b->segs[0].text.len * 6144 > 4 * total)
                chunk_len = 4096;
        }
    }
    if (chunk_len) {
        agc_hip_ctx::ChunkBufs &cb = c->chunk_bufs[lane];
        std::vector<ChunkJob> jobs;
        std::vector<uint32_t> seg0(n + 1);
        for (uint32_t i = 0; i < n; ++i) {
            seg0[i] = (uint32_t)jobs.size();
            const uint32_t len = (uint32_t)b->segs[i].text.len, nc = std::max<uint32_t>(1, (len + chunk_len - 1) / chunk_len);
            for (uint32_t ch = 0; ch < nc; ++ch)
                jobs.push_back({i, ch});
        }
        seg0[n] = (uint32_t)jobs.size();
        if (jobs.size() <= (1u << 22)) {
            ChunkPlan pl;
            pl.n_jobs = (uint32_t)jobs.size();
            pl.chunk_len = chunk_len;
            pl.cap = chunk_len / 8 + 4;
            pl.chunk_stride = MODE == MODE_ENCODE ? ((chunk_len + 5 * chunk_len / 16 + 160 + 15) & ~15u) : 0;
            CHK(ensure(c, cb.d_jobs, jobs.size() * sizeof(ChunkJob), st));
            CHK(ensure(c, cb.d_seg0, seg0.size() * 4, st));
            if (cb.logs_ahead.valid()) { // (allocated ahead by a helper thread: see ChunkBufs)
                const DevBuf ahead = cb.logs_ahead.get();
                if (ahead.p && ahead.cap > cb.d_logs.cap) {
                    if (cb.d_logs.p) {
                        HIPCHK(c, hipStreamSynchronize(st));
                        HIPCHK(c, hipFree(cb.d_logs.p));
                    }
                    cb.d_logs = ahead;
                } else if (ahead.p)
                    (void)hipFree(ahead.p);
            }
            CHK(ensure(c, cb.d_logs, jobs.size() * (size_t)pl.cap * sizeof(ChunkState), st));
            CHK(ensure(c, cb.d_logn, jobs.size() * 4, st));
            if (MODE == MODE_ENCODE)
                CHK(ensure(c, cb.d_out, jobs.size() * (size_t)pl.chunk_stride + 64, st));
            // (the two lists go through the pinned ring in pieces: a window of diverged genomes has a few 10 k chunks)
            for (size_t o = 0; o < jobs.size(); o += 1u << 19)
                CHK(upload(c, (ChunkJob *)cb.d_jobs.p + o, jobs.data() + o, std::min<size_t>(jobs.size() - o, 1u << 19) * sizeof(ChunkJob), st));
            for (size_t o = 0; o < seg0.size(); o += 1u << 20)
                CHK(upload(c, (uint32_t *)cb.d_seg0.p + o, seg0.data() + o, std::min<size_t>(seg0.size() - o, 1u << 20) * 4, st));
            pl.jobs = (const ChunkJob *)cb.d_jobs.p;
            pl.seg_chunk0 = (const uint32_t *)cb.d_seg0.p;
            pl.logs = (ChunkState *)cb.d_logs.p;
            pl.log_n = (uint32_t *)cb.d_logn.p;
            pl.chunk_out = (uint8_t *)cb.d_out.p;
            static const bool chunk_log = getenv("AGC_HIP_CHUNK_LOG") != nullptr; // (a measuring aid: every chunked launch on stderr)
            hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
            if (chunk_log)
                for (auto &e : ev)
                    (void)hipEventCreate(&e);
            {
                KTimer t(c, lane ? -1 : MODE == MODE_ENCODE ? AGC_HIP_K_ENCODE : MODE == MODE_ESTIMATE ? AGC_HIP_K_ESTIMATE : AGC_HIP_K_COSTVEC);
                if (chunk_log)
                    (void)hipEventRecord(ev[0], st);
                hipLaunchKernelGGL(lz_chunk_kernel<MODE>, dim3((pl.n_jobs + 3) / 4), dim3(256), 0, st, d_refs, d_segs, pl, out_u32);
                if (chunk_log)
                    (void)hipEventRecord(ev[1], st);
                hipLaunchKernelGGL(lz_hop_kernel<MODE>, dim3(grid), dim3(256), 0, st, d_refs, d_segs, n, pl, out_bytes, out_u32, d_resv, d_resp);
                if (chunk_log)
                    (void)hipEventRecord(ev[2], st);
            }
            HIPCHK(c, hipGetLastError());
            if (chunk_log) {
                (void)hipEventSynchronize(ev[2]);
                float a = 0, b2 = 0;
                (void)hipEventElapsedTime(&a, ev[0], ev[1]);
                (void)hipEventElapsedTime(&b2, ev[1], ev[2]);
                uint64_t total = 0;
                for (const SegDesc &sd : b->segs)
                    total += sd.text.len;
                fprintf(stderr, "    chunked launch mode %d: %u texts, %.2f Mb, longest %u, %u chunks of %u: chunk kernel %.3f ms, hop kernel %.3f ms\n", MODE, n, total / 1e6,
                        (uint32_t)b->segs[0].text.len, pl.n_jobs, chunk_len, a, b2);
                for (auto &e : ev)
                    (void)hipEventDestroy(e);
                if (MODE != MODE_ESTIMATE) { // (the hop parser left its counters where an estimate's peak goes)
                    std::vector<uint32_t> dbg(n);
                    (void)hipMemcpy(dbg.data(), d_resp, (size_t)n * 4, hipMemcpyDeviceToHost);
                    uint32_t shown = 0;
                    for (uint32_t i = 0; i < n && shown < 16; ++i) {
                        const uint32_t nch = ((uint32_t)b->segs[i].text.len + chunk_len - 1) / chunk_len, hops = dbg[b->segs[i].idx] >> 16;
                        if (i >= 4 && (hops * 2 >= nch || nch < 4))
                            continue; // (the first few, and every text the hop wavefront had to parse mostly by itself)
                        ++shown;
                        fprintf(stderr, "        text %u: %u symbols (rc %u), %u chunks, ref %u (gid %u): %u chunks taken over, %u matches parsed by the hop wavefront\n", i,
                                (uint32_t)b->segs[i].text.len, (uint32_t)b->segs[i].text.rc, nch, c->refs[b->segs[i].ref_slot].ref_size, b->segs[i].ref_slot, hops,
                                dbg[b->segs[i].idx] & 0xFFFFu);
                    }
                }
            }
            return AGC_HIP_OK;
        }
    }
    if (b && !n_dev && MODE != MODE_ENCODE && forced != 0 && n >= 1024) {
        // a big batch that was not worth chunks: the next one may be -- have the logs ready (sized for this batch's text + a half)
        agc_hip_ctx::ChunkBufs &cb = c->chunk_bufs[lane];
        if (!cb.logs_asked) {
            uint64_t total = 0;
            for (const SegDesc &sd : b->segs)
                total += sd.text.len;
            const size_t n_jobs = (size_t)(total / 4096 + n), want = (n_jobs + n_jobs / 2) * (size_t)(4096 / 8 + 4) * sizeof(ChunkState);
            if (total >= (32u << 20) && cb.d_logs.cap < want) { // (the few-texts launches of every step leave a small buffer behind)
                cb.logs_asked = true;
                const int dev = c->device;
                cb.logs_ahead = std::async(std::launch::async, [want, dev] {
                    DevBuf d;
                    if (hipSetDevice(dev) == hipSuccess && hipMalloc(&d.p, want) == hipSuccess)
                        d.cap = want;
                    else
                        d.p = nullptr;
                    return d;
                });
            }
        }
    }
    static const bool chunk_log2 = getenv("AGC_HIP_CHUNK_LOG") != nullptr;
    hipEvent_t pe[2] = {nullptr, nullptr};
    if (chunk_log2 && b)
        for (auto &e : pe)
            (void)hipEventCreate(&e);
    {
        KTimer t(c, lane ? -1 : MODE == MODE_ENCODE ? AGC_HIP_K_ENCODE : MODE == MODE_ESTIMATE ? AGC_HIP_K_ESTIMATE : AGC_HIP_K_COSTVEC);
        if (pe[0])
            (void)hipEventRecord(pe[0], st);
        hipLaunchKernelGGL(lz_parse_kernel<MODE>, dim3(grid), dim3(256), 0, st, d_refs, d_segs, n, out_bytes, out_u32, d_resv, d_resp, n_dev);
        if (pe[0])
            (void)hipEventRecord(pe[1], st);
    }
    if (pe[0]) {
        (void)hipEventSynchronize(pe[1]);
        float a = 0;
        (void)hipEventElapsedTime(&a, pe[0], pe[1]);
        uint64_t total = 0;
        for (const SegDesc &sd : b->segs)
            total += sd.text.len;
        fprintf(stderr, "    whole-text launch mode %d: %u texts, %.2f Mb, longest %u: %.3f ms\n", MODE, n, total / 1e6, n ? (uint32_t)b->segs[0].text.len : 0u, a);
        for (auto &e : pe)
            (void)hipEventDestroy(e);
    }
    HIPCHK(c, hipGetLastError());
    return AGC_HIP_OK;
}

int stage_host_texts(agc_hip_ctx *c, uint32_t n, const uint8_t *h_text, const uint64_t *h_off, const uint32_t *h_len,
                     std::vector<uint64_t> &doff)
{
    doff.resize(n);
    size_t tot = 0;
    for (uint32_t i = 0; i < n; ++i) {
        doff[i] = tot;
        tot += ((size_t)h_len[i] + 15) & ~(size_t)15;
    }
    CHK(ensure(c, c->d_in, tot + 64));
    for (uint32_t i = 0; i < n; ++i)
        if (h_len[i])
            HIPCHK(c, hipMemcpyAsync((uint8_t *)c->d_in.p + doff[i], h_text + h_off[i], h_len[i], hipMemcpyHostToDevice, c->stream));
    return AGC_HIP_OK;
}

} // namespace

extern "C" {

} // extern "C"

static int lz_encode_impl(agc_hip_ctx *c, uint32_t n, const uint32_t *h_gid, const PackedSrc &src, const uint64_t *h_off, const uint32_t *h_len,
                          const uint8_t *h_rc, uint8_t *h_enc, uint64_t enc_cap, uint64_t *h_enc_off)
{
    Batch b;
    CHK(prepare_batch(c, MODE_ENCODE, n, h_gid, src, h_off, h_len, h_rc, nullptr, b));
    CHK(ensure(c, c->d_scratch, b.out_total + 64));
    CHK(launch_parse<MODE_ENCODE>(c, n, (uint8_t *)c->d_scratch.p, nullptr, 0, nullptr, &b));
    std::vector<uint32_t> lens(n);
    HIPCHK(c, hipMemcpyAsync(lens.data(), c->d_resv.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (uint32_t i = 0; i < n; ++i)
        h_enc_off[i + 1] = h_enc_off[i] + lens[i];
    const uint64_t tot = h_enc_off[n];
    if (tot > enc_cap)
        return AGC_HIP_ECAP;
    if (!tot)
        return AGC_HIP_OK;
    if (!h_enc)
        return AGC_HIP_EINVAL;
    CHK(ensure(c, c->d_dstoff, (size_t)n * 8));
    CHK(ensure(c, c->d_compact, tot));
    CHK(upload(c, c->d_dstoff.p, h_enc_off, (size_t)n * 8, c->stream));
    hipLaunchKernelGGL(gather_bytes_kernel, dim3(grid_for(n, 1, 8192)), dim3(256), 0, c->stream, (const uint8_t *)c->d_scratch.p,
                       (const SegDesc *)c->d_segs.p, (const uint32_t *)c->d_resv.p, (const uint64_t *)c->d_dstoff.p, n,
                       (uint8_t *)c->d_compact.p);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(h_enc, c->d_compact.p, tot, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return AGC_HIP_OK;
}

// The encode in two halves (include/agc_hip.h): begin queues the parse and the copy of the delta lengths on the context's second
// stream and returns; end waits, lays the deltas out back to back and brings them over.  Between the two the caller may use every
// other entry point (they run on the first stream with their own scratch).
static int lz_encode_begin_impl(agc_hip_ctx *c, int lane, uint32_t n, const uint32_t *h_gid, const PackedSrc &src, const uint64_t *h_off,
                                const uint32_t *h_len, const uint8_t *h_rc)
{
    agc_hip_ctx::Lane2 &L = c->lane(lane);
    static thread_local Batch b; // (its descriptors are read by an asynchronous upload: they outlive this call)
    CHK(prepare_batch(c, MODE_ENCODE, n, h_gid, src, h_off, h_len, h_rc, nullptr, b, lane));
    CHK(ensure(c, L.d_scratch, b.out_total + 64, L.s));
    if (L.h_lens_cap < n) {
        if (L.h_lens)
            HIPCHK(c, hipHostFree(L.h_lens));
        L.h_lens = nullptr;
        L.h_lens_cap = 0;
        HIPCHK(c, hipHostMalloc((void **)&L.h_lens, ((size_t)n + n / 4 + 1024) * 4, hipHostMallocDefault));
        L.h_lens_cap = (size_t)n + n / 4 + 1024;
    }
    L.timed = c->timing;
    if (L.timed)
        (void)hipEventRecord(L.e0, L.s);
    CHK(launch_parse<MODE_ENCODE>(c, n, (uint8_t *)L.d_scratch.p, nullptr, lane, nullptr, &b));
    if (L.timed)
        (void)hipEventRecord(L.e1, L.s);
    HIPCHK(c, hipEventRecord(L.done, L.s));
    L.done_valid = true;
    HIPCHK(c, hipMemcpyAsync(L.h_lens, L.d_resv.p, (size_t)n * 4, hipMemcpyDeviceToHost, L.s));
    L.pending = true;
    return AGC_HIP_OK;
}

// common start of the two begin entry points: an abandoned encode is dropped, the first stream's work so far comes first
static int lz_encode_begin_enter(agc_hip_ctx *c, int lane, uint32_t n)
{
    agc_hip_ctx::Lane2 &L = c->lane(lane);
    if (L.pending) { // (an abandoned encode: a caller that failed between the two halves) -- dropped
        HIPCHK(c, hipStreamSynchronize(L.s));
        L.pending = false;
    }
    L.n = n;
    if (!n) {
        L.pending = true;
        return 1; // nothing to launch
    }
    HIPCHK(c, hipSetDevice(c->device));
    // the group descriptors of references registered a moment ago travel on the first stream: their copy has to be queued BEFORE the
    // event the lane waits for, or the lane's parse could read a table the copy has not reached yet (prepare_batch would queue it
    // behind the event)
    CHK(upload_refs(c));
    // everything queued on the first stream so far (index builds, a sample being packed) comes first
    HIPCHK(c, hipEventRecord(L.ready, c->stream));
    HIPCHK(c, hipStreamWaitEvent(L.s, L.ready, 0));
    return AGC_HIP_OK;
}

extern "C" {

int agc_hip_lz_encode_batch_packed(agc_hip_ctx *c, uint32_t n, const uint32_t *h_gid, const agc_hip_packed *pk, const uint64_t *h_off,
                                   const uint32_t *h_len, const uint8_t *h_rc, uint8_t *h_enc, uint64_t enc_cap, uint64_t *h_enc_off)
{
    if (!c || !h_enc_off || (n && (!h_gid || !h_off || !h_len || !pk || !pk->d_words)))
        return AGC_HIP_EINVAL;
    h_enc_off[0] = 0;
    if (!n)
        return AGC_HIP_OK;
    HIPCHK(c, hipSetDevice(c->device));
    return lz_encode_impl(c, n, h_gid, src_of(pk), h_off, h_len, h_rc, h_enc, enc_cap, h_enc_off);
}

int agc_hip_lz_encode_batch_dev(agc_hip_ctx *c, uint32_t n, const uint32_t *h_gid, const uint8_t *d_base, const uint64_t *h_off,
                                const uint32_t *h_len, const uint8_t *h_rc, uint8_t *h_enc, uint64_t enc_cap, uint64_t *h_enc_off)
{
    if (!c || !h_enc_off || (n && (!h_gid || !h_off || !h_len || !d_base)))
        return AGC_HIP_EINVAL;
    h_enc_off[0] = 0;
    if (!n)
        return AGC_HIP_OK;
    HIPCHK(c, hipSetDevice(c->device));
    PackedSrc src;
    std::vector<uint64_t> off2;
    CHK(pack_range(c, d_base, n, h_off, h_len, c->pk1, c->stream, src, off2));
    return lz_encode_impl(c, n, h_gid, src, off2.data(), h_len, h_rc, h_enc, enc_cap, h_enc_off);
}

int agc_hip_lz_encode_begin_packed_on(agc_hip_ctx *c, uint32_t lane, uint32_t n, const uint32_t *h_gid, const agc_hip_packed *pk, const uint64_t *h_off,
                                      const uint32_t *h_len, const uint8_t *h_rc)
{
    if (!c || lane >= AGC_HIP_ENCODE_LANES || (n && (!h_gid || !h_off || !h_len || !pk || !pk->d_words)))
        return AGC_HIP_EINVAL;
    const int e = lz_encode_begin_enter(c, (int)lane + 1, n);
    if (e)
        return e > 0 ? AGC_HIP_OK : e;
    return lz_encode_begin_impl(c, (int)lane + 1, n, h_gid, src_of(pk), h_off, h_len, h_rc);
}

int agc_hip_lz_encode_begin_packed(agc_hip_ctx *c, uint32_t n, const uint32_t *h_gid, const agc_hip_packed *pk, const uint64_t *h_off,
                                   const uint32_t *h_len, const uint8_t *h_rc)
{
    return agc_hip_lz_encode_begin_packed_on(c, 0, n, h_gid, pk, h_off, h_len, h_rc);
}

int agc_hip_lz_encode_begin_dev(agc_hip_ctx *c, uint32_t n, const uint32_t *h_gid, const uint8_t *d_base, const uint64_t *h_off,
                                const uint32_t *h_len, const uint8_t *h_rc)
{
    if (!c || (n && (!h_gid || !h_off || !h_len || !d_base)))
        return AGC_HIP_EINVAL;
    const int e = lz_encode_begin_enter(c, 1, n);
    if (e)
        return e > 0 ? AGC_HIP_OK : e;
    agc_hip_ctx::Lane2 &L = c->lane(1);
    PackedSrc src;
    std::vector<uint64_t> off2;
    CHK(pack_range(c, d_base, n, h_off, h_len, L.pk, L.s, src, off2)); // (the lane's own packed copy: it lives until end)
    return lz_encode_begin_impl(c, 1, n, h_gid, src, off2.data(), h_len, h_rc);
}

int agc_hip_lz_encode_pending(agc_hip_ctx *c, uint32_t *h_n) { return agc_hip_lz_encode_pending_on(c, 0, h_n); }

int agc_hip_lz_encode_pending_on(agc_hip_ctx *c, uint32_t lane, uint32_t *h_n)
{
    if (!c || !h_n || lane >= AGC_HIP_ENCODE_LANES)
        return AGC_HIP_EINVAL;
    agc_hip_ctx::Lane2 &L = c->lane((int)lane + 1);
    *h_n = 0;
    if (!L.pending)
        return AGC_HIP_OK;
    if (L.n_pinned) { // (launched from descriptors made on the device: the count was copied behind the parse)
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, hipStreamSynchronize(L.s));
        L.n = *L.n_pinned;
        L.n_pinned = nullptr;
    }
    *h_n = L.n;
    return AGC_HIP_OK;
}

int agc_hip_lz_encode_drop_on(agc_hip_ctx *c, uint32_t lane)
{
    if (!c || lane >= AGC_HIP_ENCODE_LANES)
        return AGC_HIP_EINVAL;
    agc_hip_ctx::Lane2 &L = c->lane((int)lane + 1);
    if (L.pending) {
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, hipStreamSynchronize(L.s));
    }
    L.pending = false;
    L.n_pinned = nullptr;
    L.timed = false;
    L.n = 0;
    return AGC_HIP_OK;
}

int agc_hip_lz_encode_end(agc_hip_ctx *c, uint8_t *h_enc, uint64_t enc_cap, uint64_t *h_enc_off)
{
    return agc_hip_lz_encode_end_on(c, 0, h_enc, enc_cap, h_enc_off);
}

int agc_hip_lz_encode_end_on(agc_hip_ctx *c, uint32_t lane, uint8_t *h_enc, uint64_t enc_cap, uint64_t *h_enc_off)
{
    if (!c || !h_enc_off || lane >= AGC_HIP_ENCODE_LANES)
        return AGC_HIP_EINVAL;
    agc_hip_ctx::Lane2 &L = c->lane((int)lane + 1);
    if (!L.pending) {
        c->err = "lz_encode_end: no encode in flight";
        return AGC_HIP_EINVAL;
    }
    h_enc_off[0] = 0;
    if (L.n_pinned) { // (launched by agc_hip_segments_encode_known: the count was copied behind the parse)
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, hipStreamSynchronize(L.s));
        L.n = *L.n_pinned;
        L.n_pinned = nullptr;
    }
    const uint32_t n = L.n;
    if (!n) {
        L.pending = false;
        return AGC_HIP_OK;
    }
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(L.s));
    if (L.timed) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, L.e0, L.e1) == hipSuccess) {
            c->ms[AGC_HIP_K_ENCODE] += ms;
            c->launches[AGC_HIP_K_ENCODE] += 1;
        }
        L.timed = false; // (a second call after AGC_HIP_ECAP must not count the launch twice)
    }
#ifdef AGC_PHASES
    {
        unsigned long long acc[8], cnt[8];
        (void)hipMemcpyFromSymbol(acc, HIP_SYMBOL(agc::g_phase_acc), sizeof acc);
        (void)hipMemcpyFromSymbol(cnt, HIP_SYMBOL(agc::g_phase_cnt), sizeof cnt);
        static const char *nm[5] = {"emit", "window", "probe", "literals", "verify"};
        unsigned long long tot = 0;
        for (int k = 0; k < 5; ++k)
            tot += acc[k];
        for (int k = 0; k < 5; ++k)
            fprintf(stderr, "phase %-8s %6.2f %%  %.0f cycles per visit (%llu visits)\n", nm[k], 100.0 * acc[k] / (tot ? tot : 1), cnt[k] ? (double)acc[k] / cnt[k] : 0.0, cnt[k]);
        fprintf(stderr, "phase total %.0f cycles per wave (%u waves)\n", (double)tot / n, n);
        (void)hipMemset((void *)nullptr, 0, 0);
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(agc::g_phase_acc), z, sizeof z);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(agc::g_phase_cnt), z, sizeof z);
    }
#endif
    static const bool laps = getenv("AGC_HIP_LAPS") != nullptr;
    auto tnow = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double lt = laps ? tnow() : 0;
    auto LAP = [&](const char *what) {
        if (laps && n > 10000) {
            const double t = tnow();
            fprintf(stderr, "    lz_encode_end lap %s %.3f ms\n", what, t - lt);
            lt = t;
        }
    };
    for (uint32_t i = 0; i < n; ++i)
        h_enc_off[i + 1] = h_enc_off[i] + L.h_lens[i];
    const uint64_t tot = h_enc_off[n];
    if (tot > enc_cap)
        return AGC_HIP_ECAP; // (still in flight: call again with a larger buffer)
    LAP("offsets");
    if (tot) {
        if (!h_enc)
            return AGC_HIP_EINVAL;
        CHK(ensure(c, L.d_dstoff, (size_t)n * 8, L.s));
        CHK(upload(c, L.d_dstoff.p, h_enc_off, (size_t)n * 8, L.s));
        // a pinned result buffer (agc_hip_host_alloc) is written by the gather itself, over the link: no second buffer, no copy
        // engine; anything else gets the deltas compacted in HBM and copied
        void *d_host = nullptr;
        if (hipHostGetDevicePointer(&d_host, h_enc, 0) != hipSuccess) {
            (void)hipGetLastError();
            d_host = nullptr;
        }
        if (!d_host)
            CHK(ensure(c, L.d_compact, tot, L.s));
        hipLaunchKernelGGL(gather_bytes_kernel, dim3(grid_for(n, 1, 8192)), dim3(256), 0, L.s, (const uint8_t *)L.d_scratch.p,
                           (const SegDesc *)L.d_segs.p, (const uint32_t *)L.d_resv.p, (const uint64_t *)L.d_dstoff.p, n,
                           d_host ? (uint8_t *)d_host : (uint8_t *)L.d_compact.p);
        HIPCHK(c, hipGetLastError());
        if (!d_host)
            HIPCHK(c, hipMemcpyAsync(h_enc, L.d_compact.p, tot, hipMemcpyDeviceToHost, L.s));
        LAP("queued");
        HIPCHK(c, hipStreamSynchronize(L.s));
        LAP("deltas on the host");
    }
    L.pending = false;
    return AGC_HIP_OK;
}

// pinned host memory for the buffers results are copied into (a copy into pageable memory goes through a bounce buffer)
int agc_hip_host_alloc(agc_hip_ctx *c, uint64_t bytes, void **out)
{
    if (!c || !out)
        return AGC_HIP_EINVAL;
    *out = nullptr;
    HIPCHK(c, hipSetDevice(c->device));
    void *p = nullptr;
    HIPCHK(c, hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault));
    {
        std::lock_guard<std::mutex> lk(c->host_alloc_mtx);
        c->host_allocs.push_back(p);
    }
    *out = p;
    return AGC_HIP_OK;
}

int agc_hip_host_free(agc_hip_ctx *c, void *p)
{
    if (!c)
        return AGC_HIP_EINVAL;
    if (!p)
        return AGC_HIP_OK;
    {
        std::lock_guard<std::mutex> lk(c->host_alloc_mtx);
        auto it = std::find(c->host_allocs.begin(), c->host_allocs.end(), p);
        if (it == c->host_allocs.end())
            return AGC_HIP_EINVAL;
        c->host_allocs.erase(it);
    }
    if (hipHostFree(p) != hipSuccess) // (no write to the shared error string: this may be the thread that collects an encode)
        return AGC_HIP_ENODEV;
    return AGC_HIP_OK;
}

} // extern "C"

static int lz_estimate_impl(agc_hip_ctx *c, uint32_t n, const uint32_t *h_gid, const PackedSrc &src, const uint64_t *h_off, const uint32_t *h_len,
                            const uint8_t *h_rc, uint32_t *h_cost, uint32_t *h_peak)
{
    Batch b;
    CHK(prepare_batch(c, MODE_ESTIMATE, n, h_gid, src, h_off, h_len, h_rc, nullptr, b));
    CHK(launch_parse<MODE_ESTIMATE>(c, n, nullptr, nullptr, 0, nullptr, &b));
    HIPCHK(c, hipMemcpyAsync(h_cost, c->d_resv.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    if (h_peak)
        HIPCHK(c, hipMemcpyAsync(h_peak, c->d_resp.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return AGC_HIP_OK;
}

static int lz_cost_vector_impl(agc_hip_ctx *c, uint32_t n, const uint32_t *h_gid, const PackedSrc &src, const uint64_t *h_off, const uint32_t *h_len,
                               const uint8_t *h_rc, const uint8_t *h_prefix_costs, uint32_t *h_costs)
{
    Batch b;
    CHK(prepare_batch(c, MODE_COSTVEC, n, h_gid, src, h_off, h_len, h_rc, h_prefix_costs, b));
    CHK(ensure(c, c->d_scratch, b.out_total + 64)); // (one byte per position on the device: lz_kernels.hip, cost_t)
    CHK(launch_parse<MODE_COSTVEC>(c, n, nullptr, (uint32_t *)c->d_scratch.p, 0, nullptr, &b));
    std::vector<uint8_t> tmp(b.out_total);
    if (b.out_total)
        HIPCHK(c, hipMemcpyAsync(tmp.data(), c->d_scratch.p, b.out_total, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (uint64_t i = 0; i < b.out_total; ++i)
        h_costs[i] = tmp[i];
    return AGC_HIP_OK;
}

static int lz_split_point_impl(agc_hip_ctx *c, uint32_t n, const uint32_t *h_gid1, const uint32_t *h_gid2, const PackedSrc &src, const uint64_t *h_off,
                               const uint32_t *h_len, const uint8_t *h_rc1, const uint8_t *h_prefix1, const uint8_t *h_rc2, const uint8_t *h_prefix2,
                               uint32_t *h_best_pos, uint32_t *h_best_sum)
{
    // 2n cost-vector parses: job 2s = (gid1, rc1, prefix1), job 2s+1 = (gid2, rc2, prefix2)
    const uint32_t m = 2 * n;
    std::vector<uint32_t> gid(m), len(m);
    std::vector<uint64_t> off(m);
    std::vector<uint8_t> rc(m), pf(m);
    for (uint32_t s = 0; s < n; ++s) {
        gid[2 * s] = h_gid1[s];
        gid[2 * s + 1] = h_gid2[s];
        off[2 * s] = off[2 * s + 1] = h_off[s];
        len[2 * s] = len[2 * s + 1] = h_len[s];
        rc[2 * s] = h_rc1[s];
        rc[2 * s + 1] = h_rc2[s];
        pf[2 * s] = h_prefix1[s];
        pf[2 * s + 1] = h_prefix2[s];
    }
    Batch b;
    CHK(prepare_batch(c, MODE_COSTVEC, m, gid.data(), src, off.data(), len.data(), rc.data(), pf.data(), b));
    CHK(ensure(c, c->d_scratch, b.out_total + 64));
    CHK(launch_parse<MODE_COSTVEC>(c, m, nullptr, (uint32_t *)c->d_scratch.p, 0, nullptr, &b));
    std::vector<SplitJob> jobs(n);
    uint64_t o = 0;
    for (uint32_t s = 0; s < n; ++s) {
        jobs[s].off1 = o;
        jobs[s].off2 = o + h_len[s];
        jobs[s].n = h_len[s];
        jobs[s].rev1 = h_prefix1[s] ? 0u : 1u;
        jobs[s].rev2 = h_prefix2[s] ? 1u : 0u;
        jobs[s].pad = 0;
        o += 2ULL * h_len[s];
    }
    CHK(ensure(c, c->d_jobs, (size_t)n * sizeof(SplitJob)));
    CHK(ensure(c, c->d_dstoff, (size_t)n * 8));
    CHK(upload(c, c->d_jobs.p, jobs.data(), (size_t)n * sizeof(SplitJob), c->stream));
    uint32_t *d_pos = (uint32_t *)c->d_dstoff.p, *d_sum = d_pos + n;
    {
        KTimer t(c, AGC_HIP_K_COSTVEC);
        hipLaunchKernelGGL(split_point_kernel, dim3(n), dim3(256), 0, c->stream, (const SplitJob *)c->d_jobs.p,
                           (const cost_t *)c->d_scratch.p, d_pos, d_sum);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(h_best_pos, d_pos, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    if (h_best_sum)
        HIPCHK(c, hipMemcpyAsync(h_best_sum, d_sum, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return AGC_HIP_OK;
}

static int fetch_slices_impl(agc_hip_ctx *c, uint32_t n, const PackedSrc &src, const uint64_t *h_off, const uint32_t *h_len, const uint8_t *h_rc,
                             uint8_t *h_out, uint64_t out_cap, uint64_t *h_out_off)
{
    for (uint32_t i = 0; i < n; ++i)
        h_out_off[i + 1] = h_out_off[i] + h_len[i];
    const uint64_t tot = h_out_off[n];
    if (tot > out_cap)
        return AGC_HIP_ECAP;
    if (!tot)
        return AGC_HIP_OK;
    if (!h_out)
        return AGC_HIP_EINVAL;
    if (!slices_inside(src, n, h_off, h_len))
        return AGC_HIP_EINVAL;
    CHK(ensure(c, c->d_compact, tot + 64));
    std::vector<ViewJob> sl(n);
    for (uint32_t i = 0; i < n; ++i)
        sl[i] = {{src.words, src.esc_index, src.esc_bytes, h_off[i], h_len[i], h_rc ? (uint32_t)(h_rc[i] != 0) : 0u}, (uint8_t *)c->d_compact.p + h_out_off[i]};
    CHK(ensure(c, c->d_slices, (size_t)n * sizeof(ViewJob)));
    CHK(upload(c, c->d_slices.p, sl.data(), (size_t)n * sizeof(ViewJob), c->stream));
    {
        KTimer t(c, AGC_HIP_K_REVCOMP);
        hipLaunchKernelGGL(slice_expand_kernel, dim3(grid_for(n, 1, 65536)), dim3(256), 0, c->stream, (const ViewJob *)c->d_slices.p, n);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(h_out, c->d_compact.p, tot, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return AGC_HIP_OK;
}

static int ref_lag_counts_impl(agc_hip_ctx *c, uint32_t n, const PackedSrc &src, const uint64_t *h_off, const uint32_t *h_len, const uint8_t *h_rc,
                               uint32_t *h_cnt, uint32_t *h_cur)
{
    if (!slices_inside(src, n, h_off, h_len))
        return AGC_HIP_EINVAL;
    std::vector<ViewJob> sl(n);
    for (uint32_t i = 0; i < n; ++i)
        sl[i] = {{src.words, src.esc_index, src.esc_bytes, h_off[i], h_len[i], h_rc ? (uint32_t)(h_rc[i] != 0) : 0u}, nullptr};
    CHK(ensure(c, c->d_slices, (size_t)n * sizeof(ViewJob)));
    CHK(ensure(c, c->d_lag, (size_t)n * 28 * 4 * 2));
    CHK(upload(c, c->d_slices.p, sl.data(), (size_t)n * sizeof(ViewJob), c->stream));
    uint32_t *d_cnt = (uint32_t *)c->d_lag.p, *d_cur = d_cnt + (size_t)n * 28;
    const uint32_t split = n >= 2048 ? 1u : std::min<uint32_t>(32u, 2048u / n);
    HIPCHK(c, hipMemsetAsync(c->d_lag.p, 0, (size_t)n * 28 * 4 * 2, c->stream));
    {
        KTimer t(c, AGC_HIP_K_REFSTORE);
        hipLaunchKernelGGL(lag_counts_kernel, dim3(n * split), dim3(256), 0, c->stream, (const ViewJob *)c->d_slices.p, d_cnt, d_cur, split);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(h_cnt, d_cnt, (size_t)n * 28 * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(h_cur, d_cur, (size_t)n * 28 * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return AGC_HIP_OK;
}

extern "C" {

// The new references (and raw items) of a registration in ONE submission that nobody on the steps' stream waits for: the
// repetitiveness counters of the first n_refs slices (agc_hip_ref_lag_counts_packed) and the symbols of all n slices
// (agc_hip_fetch_slices_packed), on a stream and in buffers of their own; _end waits for the slot (include/agc_hip.h).
int agc_hip_ref_store_begin_packed(agc_hip_ctx *c, uint32_t slot, uint32_t n_refs, uint32_t n, const agc_hip_packed *pk, const uint64_t *h_off, const uint32_t *h_len,
                                   const uint8_t *h_rc, uint32_t *h_cnt, uint32_t *h_cur, uint8_t *h_out, uint64_t out_cap, uint64_t *h_out_off)
{
    if (!c || slot >= 2 || n_refs > n || !h_out_off || (n && (!pk || !pk->d_words || !h_off || !h_len)) || (n_refs && (!h_cnt || !h_cur)))
        return AGC_HIP_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    agc_hip_ctx::RefStore &R = c->ref_store[slot];
    if (!c->ref_store_stream)
        HIPCHK(c, hipStreamCreateWithFlags(&c->ref_store_stream, hipStreamNonBlocking));
    if (!R.done)
        HIPCHK(c, hipEventCreateWithFlags(&R.done, hipEventDisableTiming));
    const hipStream_t st = c->ref_store_stream;
    if (R.pending) { // (a slot nobody collected: its copies must have landed before its buffers are written again)
        HIPCHK(c, hipEventSynchronize(R.done));
        R.pending = false;
    }
    h_out_off[0] = 0;
    for (uint32_t i = 0; i < n; ++i)
        h_out_off[i + 1] = h_out_off[i] + h_len[i];
    const uint64_t tot = h_out_off[n];
    if (tot > out_cap)
        return AGC_HIP_ECAP;
    if (!n)
        return AGC_HIP_OK;
    if (tot && !h_out)
        return AGC_HIP_EINVAL;
    const PackedSrc src = src_of(pk);
    if (!slices_inside(src, n, h_off, h_len))
        return AGC_HIP_EINVAL;
    CHK(ensure(c, R.d_out, tot + 64, st));
    CHK(ensure(c, R.d_slices, (size_t)n * sizeof(ViewJob), st));
    CHK(ensure(c, R.d_lag, std::max<size_t>(64, (size_t)n_refs * 28 * 4 * 2), st));
    std::vector<ViewJob> sl(n);
    for (uint32_t i = 0; i < n; ++i)
        sl[i] = {{src.words, src.esc_index, src.esc_bytes, h_off[i], h_len[i], h_rc ? (uint32_t)(h_rc[i] != 0) : 0u}, (uint8_t *)R.d_out.p + h_out_off[i]};
    CHK(upload(c, R.d_slices.p, sl.data(), (size_t)n * sizeof(ViewJob), st));
    if (n_refs) {
        uint32_t *d_cnt = (uint32_t *)R.d_lag.p, *d_cur = d_cnt + (size_t)n_refs * 28;
        const uint32_t split = n_refs >= 2048 ? 1u : std::min<uint32_t>(32u, 2048u / n_refs);
        HIPCHK(c, hipMemsetAsync(R.d_lag.p, 0, (size_t)n_refs * 28 * 4 * 2, st));
        hipLaunchKernelGGL(lag_counts_kernel, dim3(n_refs * split), dim3(256), 0, st, (const ViewJob *)R.d_slices.p, d_cnt, d_cur, split);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(h_cnt, d_cnt, (size_t)n_refs * 28 * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipMemcpyAsync(h_cur, d_cur, (size_t)n_refs * 28 * 4, hipMemcpyDeviceToHost, st));
    }
    if (tot) {
        hipLaunchKernelGGL(slice_expand_kernel, dim3(grid_for(n, 1, 65536)), dim3(256), 0, st, (const ViewJob *)R.d_slices.p, n);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(h_out, R.d_out.p, tot, hipMemcpyDeviceToHost, st));
    }
    HIPCHK(c, hipEventRecord(R.done, st));
    R.pending = true;
    return AGC_HIP_OK;
}

int agc_hip_ref_store_end(agc_hip_ctx *c, uint32_t slot)
{
    if (!c || slot >= 2)
        return AGC_HIP_EINVAL;
    agc_hip_ctx::RefStore &R = c->ref_store[slot];
    if (!R.pending)
        return AGC_HIP_OK;
    // (no hipSetDevice, no write to the shared error string: this may be the bookkeeping thread)
    if (hipEventSynchronize(R.done) != hipSuccess)
        return AGC_HIP_ENODEV;
    R.pending = false;
    return AGC_HIP_OK;
}

// ---- sequences in a 2-bit packed buffer -------------------------------------
int agc_hip_lz_estimate_batch_packed(agc_hip_ctx *c, uint32_t n, const uint32_t *h_gid, const agc_hip_packed *pk, const uint64_t *h_off,
                                     const uint32_t *h_len, const uint8_t *h_rc, uint32_t *h_cost, uint32_t *h_peak)
{
    if (!c || (n && (!h_gid || !h_off || !h_len || !pk || !pk->d_words || !h_cost)))
        return AGC_HIP_EINVAL;
    if (!n)
        return AGC_HIP_OK;
    HIPCHK(c, hipSetDevice(c->device));
    return lz_estimate_impl(c, n, h_gid, src_of(pk), h_off, h_len, h_rc, h_cost, h_peak);
}

int agc_hip_lz_cost_vector_batch_packed(agc_hip_ctx *c, uint32_t n, const uint32_t *h_gid, const agc_hip_packed *pk, const uint64_t *h_off,
                                        const uint32_t *h_len, const uint8_t *h_rc, const uint8_t *h_prefix_costs, uint32_t *h_costs)
{
    if (!c || (n && (!h_gid || !h_off || !h_len || !pk || !pk->d_words || !h_costs)))
        return AGC_HIP_EINVAL;
    if (!n)
        return AGC_HIP_OK;
    HIPCHK(c, hipSetDevice(c->device));
    return lz_cost_vector_impl(c, n, h_gid, src_of(pk), h_off, h_len, h_rc, h_prefix_costs, h_costs);
}

int agc_hip_lz_split_point_batch_packed(agc_hip_ctx *c, uint32_t n, const uint32_t *h_gid1, const uint32_t *h_gid2, const agc_hip_packed *pk,
                                        const uint64_t *h_off, const uint32_t *h_len, const uint8_t *h_rc1, const uint8_t *h_prefix1,
                                        const uint8_t *h_rc2, const uint8_t *h_prefix2, uint32_t *h_best_pos, uint32_t *h_best_sum)
{
    if (!c || (n && (!h_gid1 || !h_gid2 || !pk || !pk->d_words || !h_off || !h_len || !h_rc1 || !h_prefix1 || !h_rc2 || !h_prefix2 || !h_best_pos)))
        return AGC_HIP_EINVAL;
    if (!n)
        return AGC_HIP_OK;
    HIPCHK(c, hipSetDevice(c->device));
    return lz_split_point_impl(c, n, h_gid1, h_gid2, src_of(pk), h_off, h_len, h_rc1, h_prefix1, h_rc2, h_prefix2, h_best_pos, h_best_sum);
}

int agc_hip_fetch_slices_packed(agc_hip_ctx *c, uint32_t n, const agc_hip_packed *pk, const uint64_t *h_off, const uint32_t *h_len,
                                const uint8_t *h_rc, uint8_t *h_out, uint64_t out_cap, uint64_t *h_out_off)
{
    if (!c || !h_out_off || (n && (!pk || !pk->d_words || !h_off || !h_len)))
        return AGC_HIP_EINVAL;
    h_out_off[0] = 0;
    if (!n)
        return AGC_HIP_OK;
    HIPCHK(c, hipSetDevice(c->device));
    return fetch_slices_impl(c, n, src_of(pk), h_off, h_len, h_rc, h_out, out_cap, h_out_off);
}

int agc_hip_ref_lag_counts_packed(agc_hip_ctx *c, uint32_t n, const agc_hip_packed *pk, const uint64_t *h_off, const uint32_t *h_len,
                                  const uint8_t *h_rc, uint32_t *h_cnt, uint32_t *h_cur)
{
    if (!c || (n && (!pk || !pk->d_words || !h_off || !h_len || !h_cnt || !h_cur)))
        return AGC_HIP_EINVAL;
    if (!n)
        return AGC_HIP_OK;
    HIPCHK(c, hipSetDevice(c->device));
    return ref_lag_counts_impl(c, n, src_of(pk), h_off, h_len, h_rc, h_cnt, h_cur);
}

// ---- one byte per symbol (device): packed first -------------------------------
int agc_hip_lz_estimate_batch_dev(agc_hip_ctx *c, uint32_t n, const uint32_t *h_gid, const uint8_t *d_base, const uint64_t *h_off,
                                  const uint32_t *h_len, const uint8_t *h_rc, uint32_t *h_cost, uint32_t *h_peak)
{
    if (!c || (n && (!h_gid || !h_off || !h_len || !d_base || !h_cost)))
        return AGC_HIP_EINVAL;
    if (!n)
        return AGC_HIP_OK;
    HIPCHK(c, hipSetDevice(c->device));
    PackedSrc src;
    std::vector<uint64_t> off2;
    CHK(pack_range(c, d_base, n, h_off, h_len, c->pk1, c->stream, src, off2));
    return lz_estimate_impl(c, n, h_gid, src, off2.data(), h_len, h_rc, h_cost, h_peak);
}

int agc_hip_lz_cost_vector_batch_dev(agc_hip_ctx *c, uint32_t n, const uint32_t *h_gid, const uint8_t *d_base,
                                     const uint64_t *h_off, const uint32_t *h_len, const uint8_t *h_rc,
                                     const uint8_t *h_prefix_costs, uint32_t *h_costs)
{
    if (!c || (n && (!h_gid || !h_off || !h_len || !d_base || !h_costs)))
        return AGC_HIP_EINVAL;
    if (!n)
        return AGC_HIP_OK;
    HIPCHK(c, hipSetDevice(c->device));
    PackedSrc src;
    std::vector<uint64_t> off2;
    CHK(pack_range(c, d_base, n, h_off, h_len, c->pk1, c->stream, src, off2));
    return lz_cost_vector_impl(c, n, h_gid, src, off2.data(), h_len, h_rc, h_prefix_costs, h_costs);
}

// host-resident texts --------------------------------------------------------
int agc_hip_lz_encode_batch(agc_hip_ctx *c, uint32_t n, const uint32_t *h_gid, const uint8_t *h_text, const uint64_t *h_off,
                            const uint32_t *h_len, const uint8_t *h_rc, uint8_t *h_enc, uint64_t enc_cap, uint64_t *h_enc_off)
{
    if (!c || (n && (!h_text || !h_off || !h_len)))
        return AGC_HIP_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<uint64_t> doff;
    CHK(stage_host_texts(c, n, h_text, h_off, h_len, doff));
    return agc_hip_lz_encode_batch_dev(c, n, h_gid, (const uint8_t *)c->d_in.p, doff.data(), h_len, h_rc, h_enc, enc_cap, h_enc_off);
}

int agc_hip_lz_estimate_batch(agc_hip_ctx *c, uint32_t n, const uint32_t *h_gid, const uint8_t *h_text, const uint64_t *h_off,
                              const uint32_t *h_len, const uint8_t *h_rc, uint32_t *h_cost, uint32_t *h_peak)
{
    if (!c || (n && (!h_text || !h_off || !h_len)))
        return AGC_HIP_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<uint64_t> doff;
    CHK(stage_host_texts(c, n, h_text, h_off, h_len, doff));
    return agc_hip_lz_estimate_batch_dev(c, n, h_gid, (const uint8_t *)c->d_in.p, doff.data(), h_len, h_rc, h_cost, h_peak);
}

int agc_hip_lz_cost_vector_batch(agc_hip_ctx *c, uint32_t n, const uint32_t *h_gid, const uint8_t *h_text, const uint64_t *h_off,
                                 const uint32_t *h_len, const uint8_t *h_rc, const uint8_t *h_prefix_costs, uint32_t *h_costs)
{
    if (!c || (n && (!h_text || !h_off || !h_len)))
        return AGC_HIP_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<uint64_t> doff;
    CHK(stage_host_texts(c, n, h_text, h_off, h_len, doff));
    return agc_hip_lz_cost_vector_batch_dev(c, n, h_gid, (const uint8_t *)c->d_in.p, doff.data(), h_len, h_rc, h_prefix_costs,
                                            h_costs);
}

int agc_hip_lz_split_point_batch_dev(agc_hip_ctx *c, uint32_t n, const uint32_t *h_gid1, const uint32_t *h_gid2,
                                     const uint8_t *d_base, const uint64_t *h_off, const uint32_t *h_len, const uint8_t *h_rc1,
                                     const uint8_t *h_prefix1, const uint8_t *h_rc2, const uint8_t *h_prefix2, uint32_t *h_best_pos,
                                     uint32_t *h_best_sum)
{
    if (!c || (n && (!h_gid1 || !h_gid2 || !d_base || !h_off || !h_len || !h_rc1 || !h_prefix1 || !h_rc2 || !h_prefix2 || !h_best_pos)))
        return AGC_HIP_EINVAL;
    if (!n)
        return AGC_HIP_OK;
    HIPCHK(c, hipSetDevice(c->device));
    PackedSrc src;
    std::vector<uint64_t> off2;
    CHK(pack_range(c, d_base, n, h_off, h_len, c->pk1, c->stream, src, off2));
    return lz_split_point_impl(c, n, h_gid1, h_gid2, src, off2.data(), h_len, h_rc1, h_prefix1, h_rc2, h_prefix2, h_best_pos, h_best_sum);
}

int agc_hip_fetch_slices_dev(agc_hip_ctx *c, uint32_t n, const uint8_t *d_base, const uint64_t *h_off, const uint32_t *h_len,
                             const uint8_t *h_rc, uint8_t *h_out, uint64_t out_cap, uint64_t *h_out_off)
{
    if (!c || !h_out_off || (n && (!d_base || !h_off || !h_len)))
        return AGC_HIP_EINVAL;
    h_out_off[0] = 0;
    if (!n)
        return AGC_HIP_OK;
    HIPCHK(c, hipSetDevice(c->device));
    PackedSrc src;
    std::vector<uint64_t> off2;
    CHK(pack_range(c, d_base, n, h_off, h_len, c->pk1, c->stream, src, off2));
    return fetch_slices_impl(c, n, src, off2.data(), h_len, h_rc, h_out, out_cap, h_out_off);
}

// ---------------------------------------------------------------------------
// a13 helper
// ---------------------------------------------------------------------------
int agc_hip_ref_lag_counts_dev(agc_hip_ctx *c, uint32_t n, const uint8_t *d_base, const uint64_t *h_off, const uint32_t *h_len,
                               const uint8_t *h_rc, uint32_t *h_cnt, uint32_t *h_cur)
{
    if (!c || (n && (!d_base || !h_off || !h_len || !h_cnt || !h_cur)))
        return AGC_HIP_EINVAL;
    if (!n)
        return AGC_HIP_OK;
    HIPCHK(c, hipSetDevice(c->device));
    PackedSrc src;
    std::vector<uint64_t> off2;
    CHK(pack_range(c, d_base, n, h_off, h_len, c->pk1, c->stream, src, off2));
    return ref_lag_counts_impl(c, n, src, off2.data(), h_len, h_rc, h_cnt, h_cur);
}

// ---------------------------------------------------------------------------
// S3: zstd level-17 frames
// ---------------------------------------------------------------------------
uint32_t agc_hip_zstd17_max_input(void) { return zs::BLOCKSIZE_MAX; }

int agc_hip_zstd17_cparams(uint64_t src_size, uint32_t out7[7])
{
    if (!out7)
        return AGC_HIP_EINVAL;
    zs::level17Params(src_size, out7);
    return AGC_HIP_OK;
}

int agc_hip_zstd_cparams(int level, uint64_t src_size, uint32_t out7[7])
{
    if (!out7 || (level != 13 && level != 17 && level != 19))
        return AGC_HIP_EINVAL;
    zs::levelParams(level, src_size, out7);
    return AGC_HIP_OK;
}

uint32_t agc_hip_zstd17_resident_frames(agc_hip_ctx *c)
{
    if (!c)
        return 0;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || cus <= 0)
        return 0;
    // groups of 3 lanes: 21 frames per wave; 8 waves per CU (LDS: zgrp_lds_bytes(21, 3) = 20 328 of 160 KiB / 8; registers: 2 per SIMD)
    return (uint32_t)cus * 8u * 21u;
}

int agc_hip_zstd17_background(agc_hip_ctx *c, int on)
{
    if (!c)
        return AGC_HIP_EINVAL;
    c->zstd_background = on != 0;
    return AGC_HIP_OK;
}

// d_src_ext != nullptr: the inputs are in HBM already (input i = d_src_ext[h_src_off[i] .. h_src_off[i+1])); h_src is not looked at
static int zstd17_batch_impl(agc_hip_ctx *c, uint32_t n, const uint8_t *h_src, const uint8_t *d_src_ext, const uint64_t *h_src_off, uint8_t *h_dst,
                             uint64_t dst_cap, uint64_t *h_dst_off, const uint8_t *h_level = nullptr)
{
    if (!c || !h_dst_off || (n && (!h_src_off)))
        return AGC_HIP_EINVAL;
    h_dst_off[0] = 0;
    if (!n)
        return AGC_HIP_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const hipStream_t zs_ = c->zstream; // (this call shares no buffer, stream or event with the other entry points)
    const uint64_t src_total = h_src_off[n] - h_src_off[0];
    if (src_total && !h_src && !d_src_ext)
        return AGC_HIP_EINVAL;
    // per frame: parameters, workspace size; longest first so that the lanes of a wave finish together
    std::vector<ZFrameJob> jobs(n);
    std::vector<uint64_t> ws_need(n), dst_o(n);
    uint64_t dst_total = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const uint64_t len = h_src_off[i + 1] - h_src_off[i];
        if (h_src_off[i + 1] < h_src_off[i] || len > zs::BLOCKSIZE_MAX) {
            c->err = "zstd17_batch: input " + std::to_string(i) + " is larger than one block";
            return AGC_HIP_EINVAL;
        }
        const int level = h_level ? h_level[i] : 17;
        if (level != 13 && level != 17 && level != 19) {
            c->err = "zstd_batch: level " + std::to_string(level) + " (input " + std::to_string(i) + "): 13, 17 or 19";
            return AGC_HIP_EINVAL;
        }
        uint32_t p[7];
        zs::levelParams(level, len, p);
        const zs::CParams cp = {p[0], p[1], p[2], p[3], p[4], p[5], p[6]};
        ws_need[i] = zs::wsLayout(cp, (uint32_t)len).total;
        dst_o[i] = dst_total;
        dst_total += (zs::frameBound((uint32_t)len) + 15) & ~15u;
        jobs[i].src = 0;
        jobs[i].dst = 0;
        jobs[i].ws = 0;
        jobs[i].src_size = (uint32_t)len;
        jobs[i].idx = i;
        jobs[i].cp = cp;
        jobs[i].pad = 0;
    }
    std::vector<uint32_t> order(n);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return jobs[a].src_size > jobs[b].src_size; });

    if (!d_src_ext)
        CHK(ensure_z(c, c->d_zsrc, src_total + 64));
    const uint8_t *const d_src = d_src_ext ? d_src_ext + h_src_off[0] : (const uint8_t *)c->d_zsrc.p;
    CHK(ensure_z(c, c->d_zdst, dst_total + 64));
    CHK(ensure_z(c, c->d_zsize, (size_t)n * 4));
    CHK(ensure_z(c, c->d_zjobs, (size_t)n * sizeof(ZFrameJob)));
    if (src_total && !d_src_ext)
        HIPCHK(c, hipMemcpyAsync(c->d_zsrc.p, h_src + h_src_off[0], src_total, hipMemcpyHostToDevice, zs_));
    // workspace arena: as many frames per launch as the budget allows (a frame's tables must be zero at its start)
    size_t free_b = 0, total_b = 0;
    HIPCHK(c, hipMemGetInfo(&free_b, &total_b));
    // (the call may run on a background thread beside the entry points that add samples: half of what is free, at most 32 GB --
    // one resident round of 43 008 frames of 16 KiB needs 19 GB -- leaves the other streams' buffers room to grow)
    uint64_t budget = std::min<uint64_t>(32ull << 30, (uint64_t)((free_b + c->d_zws.cap) * 0.5));
    if (const char *e = getenv("AGC_HIP_ZSTD_ARENA_MB"))
        budget = std::max<uint64_t>(1, strtoull(e, nullptr, 10)) << 20;
    budget = std::max<uint64_t>(budget, ws_need[order[0]]);
    std::vector<ZFrameJob> sorted(n);
    for (uint32_t done = 0; done < n;) {
        uint64_t used = 0;
        uint32_t m = 0;
        while (done + m < n && used + ws_need[order[done + m]] <= budget) {
            used += ws_need[order[done + m]];
            ++m;
        }
        CHK(ensure_z(c, c->d_zws, used));
        // the frames of this launch: first the ones the one-lane kernel takes, then the ones a group of lanes takes
        // (zstd/zs_opt_grp.h: inputs of the btultra2 class, <= 16 KiB -- every delta pack of a collection below ~35 samples)
        uint32_t grp_g = c->zstd_background ? 0u : 3u;
        if (const char *e = getenv("AGC_HIP_ZSTD_GROUP"))
            grp_g = (uint32_t)std::min(3, std::max(0, atoi(e)));
        if (grp_g == 1)
            grp_g = 0;
        // three classes: one-lane kernel (inputs the group parser does not take: < 8 bytes; everything in background mode), groups
        // with one-word records (inputs <= 16 KiB), groups with two-word records (the rest, up to one block)
        auto cls = [&](uint32_t i) -> int {
            if (!grp_g || !zs::grpEligible(jobs[i].cp, jobs[i].src_size))
                return 0;
            return zs::grpWide(jobs[i].cp, jobs[i].src_size) ? 2 : 1;
        };
        std::vector<uint32_t> part;
        part.reserve(m);
        uint32_t m_cls[3] = {0, 0, 0};
        for (int want = 0; want < 3; ++want)
            for (uint32_t t = 0; t < m; ++t) {
                const uint32_t i = order[done + t];
                if (cls(i) == want) {
                    part.push_back(i);
                    ++m_cls[want];
                }
            }
        const uint32_t m_one = m_cls[0], m_grp = m_cls[1], m_wide = m_cls[2];
        used = 0;
        for (uint32_t t = 0; t < m; ++t) {
            const uint32_t i = part[t];
            ZFrameJob jb = jobs[i];
            jb.src = h_src_off[i] - h_src_off[0];
            jb.dst = dst_o[i];
            jb.ws = used;
            used += ws_need[i];
            sorted[done + t] = jb;
        }
        HIPCHK(c, hipMemsetAsync(c->d_zws.p, 0, used, zs_));
        HIPCHK(c, hipMemcpyAsync((ZFrameJob *)c->d_zjobs.p + done, sorted.data() + done, (size_t)m * sizeof(ZFrameJob), hipMemcpyHostToDevice,
                                 zs_));
        {
            ZTimer t(c);
            const uint32_t dbg = 0; // (the kernels' trace word: a debugging build sets it)
            const dim3 block(64);
            if (m_one) {
                // frames per wave: fewer = fewer distinct parser states per trip of the micro-step loop, but every wave of the
                // launch must be resident at once (2 per SIMD = 2048 on 256 CUs) or the launch takes two rounds.  Measured per
                // 36 k frames of 13 KB: 24 lanes 1.31 s, 32 lanes 1.35 s, 48 lanes 1.38 s, 64 lanes 1.48 s, 16 lanes (two rounds) 1.82 s
                uint32_t lanes = 64; // (background launches: as few waves as possible, the other streams' kernels need the slots)
                if (!c->zstd_background)
                    for (uint32_t cand : {24u, 32u, 48u})
                        if ((m_one + cand - 1) / cand <= 1900) {
                            lanes = cand;
                            break;
                        }
                const dim3 grid((m_one + lanes - 1) / lanes);
                const ZFrameJob *dj = (const ZFrameJob *)c->d_zjobs.p + done;
                if (c->zstd_background)
                    hipLaunchKernelGGL((zstd_frames_kernel<2, false>), grid, block, 0, zs_, dj, m_one, (uint32_t *)c->d_zsize.p, lanes,
                                       d_src, (uint8_t *)c->d_zdst.p, (uint8_t *)c->d_zws.p, dbg);
                else // (frequency tables and the first matches of a request in LDS)
                    hipLaunchKernelGGL((zstd_frames_kernel<2, true, true>), grid, block, (size_t)lanes * zs::FAST_WORDS * 4, zs_, dj, m_one,
                                       (uint32_t *)c->d_zsize.p, lanes, d_src, (uint8_t *)c->d_zdst.p, (uint8_t *)c->d_zws.p, dbg);
            }
            // both group classes in one call: side by side on two streams (a launch lasts as long as its longest frame)
            const bool both = m_grp && m_wide;
            const hipStream_t zs_w = both ? c->zstream2 : zs_;
            if (both) {
                HIPCHK(c, hipEventRecord(c->zev_a, zs_));
                HIPCHK(c, hipStreamWaitEvent(c->zstream2, c->zev_a, 0));
            }
            const uint32_t gpw = grp_g ? 64 / grp_g : 1; // groups (= frames) per wave
            auto launch_grp = [&](uint32_t first, uint32_t count, bool wide, hipStream_t st) {
                const dim3 grid((count + gpw - 1) / gpw);
                const ZFrameJob *dj = (const ZFrameJob *)c->d_zjobs.p + done + first;
                const uint32_t stride = wide ? ZGRP_REC_STRIDE_WIDE : ZGRP_REC_STRIDE;
                if (grp_g == 2)
                    hipLaunchKernelGGL((zstd_frames_grp_kernel<2, 2>), grid, block, zgrp_lds_bytes(gpw, 2, wide), st, dj, count, (uint32_t *)c->d_zsize.p, gpw,
                                       d_src, (uint8_t *)c->d_zdst.p, (uint8_t *)c->d_zws.p, dbg, stride);
                else
                    hipLaunchKernelGGL((zstd_frames_grp_kernel<3, 2>), grid, block, zgrp_lds_bytes(gpw, 3, wide), st, dj, count, (uint32_t *)c->d_zsize.p, gpw,
                                       d_src, (uint8_t *)c->d_zdst.p, (uint8_t *)c->d_zws.p, dbg, stride);
            };
            if (m_wide)
                launch_grp(m_one + m_grp, m_wide, true, zs_w);
            if (m_grp)
                launch_grp(m_one, m_grp, false, zs_);
            if (both) {
                HIPCHK(c, hipEventRecord(c->zev_b, c->zstream2));
                HIPCHK(c, hipStreamWaitEvent(zs_, c->zev_b, 0));
            }
        }
        HIPCHK(c, hipGetLastError());
        done += m;
    }
    if (c->h_zsizes_cap < n) {
        if (c->h_zsizes)
            HIPCHK(c, hipHostFree(c->h_zsizes));
        c->h_zsizes = nullptr;
        c->h_zsizes_cap = 0;
        const size_t cap = (size_t)n + n / 4 + 1024;
        HIPCHK(c, hipHostMalloc((void **)&c->h_zsizes, cap * 4, hipHostMallocDefault));
        c->h_zsizes_cap = cap;
    }
    uint32_t *sizes = c->h_zsizes;
    HIPCHK(c, hipMemcpyAsync(sizes, c->d_zsize.p, (size_t)n * 4, hipMemcpyDeviceToHost, zs_));
    // (the launch lasts hundreds of ms and the host pool compresses its share of the packs meanwhile: this thread sleeps)
    HIPCHK(c, hipEventRecord(c->zev_wait, zs_));
    HIPCHK(c, hipEventSynchronize(c->zev_wait));
    for (uint32_t i = 0; i < n; ++i)
        h_dst_off[i + 1] = h_dst_off[i] + sizes[i];
    const uint64_t tot = h_dst_off[n];
    if (tot > dst_cap)
        return AGC_HIP_ECAP;
    if (!h_dst)
        return AGC_HIP_EINVAL;
    // compact the frames and bring them back in one copy
    CHK(ensure_z(c, c->d_zout, tot + 64));
    CHK(ensure_z(c, c->d_zdstoff, (size_t)(n + 1) * 8));
    HIPCHK(c, hipMemcpyAsync(c->d_zdstoff.p, h_dst_off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, zs_));
    hipLaunchKernelGGL(zstd_gather_kernel, dim3(grid_for(n, 1, 16384)), dim3(256), 0, zs_, (const ZFrameJob *)c->d_zjobs.p, n,
                       (const uint64_t *)c->d_zdstoff.p, (const uint8_t *)c->d_zdst.p, (uint8_t *)c->d_zout.p);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(h_dst, c->d_zout.p, tot, hipMemcpyDeviceToHost, zs_));
    HIPCHK(c, hipStreamSynchronize(zs_));
    return AGC_HIP_OK;
}

int agc_hip_zstd_batch(agc_hip_ctx *c, uint32_t n, const uint8_t *h_src, const uint64_t *h_src_off, const uint8_t *h_level, uint8_t *h_dst,
                       uint64_t dst_cap, uint64_t *h_dst_off)
{
    return zstd17_batch_impl(c, n, h_src, nullptr, h_src_off, h_dst, dst_cap, h_dst_off, h_level);
}

int agc_hip_zstd17_batch(agc_hip_ctx *c, uint32_t n, const uint8_t *h_src, const uint64_t *h_src_off, uint8_t *h_dst, uint64_t dst_cap,
                         uint64_t *h_dst_off)
{
    return zstd17_batch_impl(c, n, h_src, nullptr, h_src_off, h_dst, dst_cap, h_dst_off);
}

int agc_hip_zstd17_batch_dev(agc_hip_ctx *c, uint32_t n, const uint8_t *d_src, const uint64_t *h_src_off, uint8_t *h_dst, uint64_t dst_cap,
                             uint64_t *h_dst_off)
{
    if (n && h_src_off && h_src_off[n] > h_src_off[0] && !d_src)
        return AGC_HIP_EINVAL;
    return zstd17_batch_impl(c, n, nullptr, d_src ? d_src : (const uint8_t *)1, h_src_off, h_dst, dst_cap, h_dst_off);
}

} // extern "C"

#include "splitters.hip"
#include "segments.hip"
