// two_ranks_one_process.cpp -- TEST TOOL (race checking): the multi-GPU single-archive protocol with TWO compressors in ONE
// process, so that a ThreadSanitizer build sees the writer rank's threads at work together: the thread that applies the other
// rank's records and commits its own samples, the bookkeeping thread, the entropy thread.  Same schedule as agc_amd/dist.py with
// prefetching ranks: a rank prepares its next sample as soon as it has committed one; commit = head, (publish), finish, body.
// Linked against the device stand-in (tests/devsim) -- never part of the product.
//   two_ranks_one_process <out.agc> <k> <min_match_len> <segment_size> <pack_cardinality> <adaptive 0|1> <ref.fa> <sample.fa>...
#include "../../agc_amd/csrc/host/compressor_impl.h"
#include "../../include/agc_hip.h"
using namespace agc;

struct Sample {
    std::string name;
    std::vector<std::string> ids;
    bytes_t codes;
    std::vector<uint64_t> off;
};

static bool load(const std::string &path, Sample &s)
{
    FastaReader fr;
    if (!fr.open(path))
        return false;
    std::string id;
    bytes_t ctg;
    s.off.assign(1, 0);
    while (fr.read_contig_raw(id, ctg)) {
        preprocess_raw_contig(ctg);
        s.ids.push_back(id);
        s.codes.insert(s.codes.end(), ctg.begin(), ctg.end());
        s.off.push_back(s.codes.size());
        ctg.clear();
    }
    s.codes.insert(s.codes.end(), 4096, 4); // (the kernels' read-ahead margin)
    size_t a = path.find_last_of('/');
    s.name = path.substr(a == std::string::npos ? 0 : a + 1);
    for (const char *suf : {".gz", ".fa", ".fasta", ".fna"})
        if (s.name.size() > strlen(suf) && s.name.compare(s.name.size() - strlen(suf), strlen(suf), suf) == 0)
            s.name.resize(s.name.size() - strlen(suf));
    return true;
}

int main(int argc, char **argv)
{
    if (argc < 9)
        return 2;
    const std::string out = argv[1];
    const uint32_t k = atoi(argv[2]), mml = atoi(argv[3]), seg = atoi(argv[4]), pack = atoi(argv[5]);
    const bool adaptive = atoi(argv[6]) != 0;
    std::vector<Sample> smp(argc - 7);
    for (int i = 7; i < argc; ++i)
        if (!load(argv[i], smp[i - 7]))
            return 3;
    const uint32_t W = getenv("TWO_RANKS_W") ? (uint32_t)std::max(2, atoi(getenv("TWO_RANKS_W"))) : 2u; // (more ranks than two on request)
    std::vector<CAGCCompressor> cmp(W);
    for (uint32_t r = 0; r < W; ++r) {
        if (!cmp[r].SetDistributed(r, W, 0))
            return 4;
        if (!cmp[r].Create(r == 0 ? out : std::string(), pack, k, argv[7], seg, mml, false, adaptive, 0, 4, 0.0))
            return 5;
    }
    const size_t n = smp.size();
    auto prepare = [&](uint32_t r, size_t i) { return cmp[r].PrepareSampleDevice(smp[i].name, smp[i].ids, smp[i].codes.data(), smp[i].off.data()); };
    std::vector<long> prepared(W, -1);
    for (size_t i = 0; i < n; ++i) {
        const uint32_t o = (uint32_t)(i % W);
        for (uint32_t r = 0; r < W; ++r) { // every rank prepares its next sample before it joins the commits in front of it (adaptive mode too)
            size_t nxt = i + (r + W - o) % W;
            if (prepared[r] < 0 && nxt < n) {
                if (!prepare(r, nxt))
                    return 6;
                prepared[r] = (long)nxt;
            }
        }
        if (prepared[o] < 0) {
            if (!prepare(o, i))
                return 6;
            prepared[o] = (long)i;
        }
        if (prepared[o] != (long)i)
            return 7;
        if (!cmp[o].CommitPreparedHead())
            return 8;
        size_t hn = 0;
        const uint8_t *hp = cmp[o].LastRecord(&hn);
        const std::vector<uint8_t> head(hp, hp + hn); // ("broadcast")
        if (!cmp[o].CommitPreparedFinish())
            return 9;
        prepared[o] = -1;
        size_t bn = 0;
        const uint8_t *body = cmp[o].LastRecordBody(&bn);
        for (uint32_t r = 0; r < W; ++r) {
            if (r == o)
                continue;
            const uint8_t *bp = nullptr;
            size_t bsz = 0;
            if (r == 0 && bn) { // the writer receives the body into the buffer its bookkeeping will read
                uint8_t *dst = cmp[r].RecordBodyBuffer(bn);
                if (!dst)
                    return 10;
                memcpy(dst, body, bn);
                bp = dst;
                bsz = bn;
            }
            if (!cmp[r].ApplyRecord(head.data(), head.size(), nullptr, bp, bsz))
                return 11;
        }
    }
    // Close as agc_amd/dist.py does: the writer's pending packs through the (stand-in) device entropy stage
    const uint8_t *src = nullptr;
    const uint64_t *off = nullptr;
    uint32_t np = 0;
    if (!cmp[0].CloseCollectPacks(&src, &off, &np))
        return 12;
    std::vector<uint8_t> frames;
    std::vector<uint64_t> foff(np + 1, 0);
    if (np) {
        frames.resize(off[np] + 64ull * np + 1024);
        agc_hip_ctx *ctx = nullptr;
        if (agc_hip_create(&ctx, 0) != AGC_HIP_OK)
            return 13;
        if (agc_hip_zstd17_batch(ctx, np, src, off, frames.data(), frames.size(), foff.data()) != AGC_HIP_OK)
            return 14;
        agc_hip_destroy(ctx);
    }
    if (!cmp[0].CloseProvideFrames(frames.data(), foff.data()))
        return 15;
    for (uint32_t r = 0; r < W; ++r)
        if (!cmp[r].Close(4))
            return 16;
    return 0;
}
