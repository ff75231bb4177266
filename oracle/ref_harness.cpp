// ref_harness.cpp -- thin extern "C" shim around the REFERENCE's own classes,
// compiled in place from /root/reference (never copied) into oracle/_ref/.
// TEST INFRASTRUCTURE ONLY: used to pin oracle/agc_oracle.c and to generate
// the golden vectors under tests/golden/.  It cannot exist on the GPU box
// unless oracle/_ref/libagcref.so was prebuilt here.
//
// Wraps: CLZDiff_V2 (src/common/lz_diff.h:375-436), CKmer (src/core/kmer.h),
// hash_set_lp (src/core/hs.h), bloom_set_t (src/core/utils_adv.h:180-282).

#include <cstdint>
#include <cstring>
#include <vector>
#include <string>

#include "src/common/lz_diff.h"
#include "src/core/kmer.h"
#include "src/core/hs.h"
#include "src/core/utils_adv.h"

extern "C" {

void *ref_lz_create(const uint8_t *ref, uint32_t n, uint32_t min_match_len)
{
    auto *z = new CLZDiff_V2(min_match_len);
    contig_t r(ref, ref + n);
    z->Prepare(r);
    return z;
}

void ref_lz_free(void *h) { delete (CLZDiff_V2 *)h; }

size_t ref_lz_encode(void *h, const uint8_t *text, uint32_t n, uint8_t *out, size_t cap)
{
    contig_t t(text, text + n), e;
    ((CLZDiff_V2 *)h)->Encode(t, e);
    if (e.size() <= cap)
        memcpy(out, e.data(), e.size());
    return e.size();
}

uint64_t ref_lz_estimate(void *h, const uint8_t *text, uint32_t n, uint32_t bound)
{
    contig_t t(text, text + n);
    return ((CLZDiff_V2 *)h)->Estimate(t, bound);
}

void ref_lz_cost_vector(void *h, const uint8_t *text, uint32_t n, int prefix, uint32_t *costs)
{
    contig_t t(text, text + n);
    std::vector<uint32_t> v;
    ((CLZDiff_V2 *)h)->AssureIndex();
    ((CLZDiff_V2 *)h)->GetCodingCostVector(t, v, prefix != 0);
    memcpy(costs, v.data(), v.size() * sizeof(uint32_t));
}

size_t ref_lz_decode(uint32_t min_match_len, const uint8_t *ref, uint32_t rn,
                     const uint8_t *enc, size_t en, uint8_t *out, size_t cap)
{
    CLZDiff_V2 z(min_match_len);
    contig_t r(ref, ref + rn), e(enc, enc + en), d;
    z.Decode(r, e, d);
    if (d.size() <= cap)
        memcpy(out, d.data(), d.size());
    return d.size();
}

// The scan loop body of CAGCCompressor::compress_contig
// (src/core/agc_compressor.cpp:2007-2036) driven with the reference's own
// CKmer, bloom_set_t and hash_set_lp (constructed as agc_compressor.h:625 and
// agc_compressor.cpp:543-555 do).  Emits (pos, kmer_dir, kmer_rc) per accepted hit.
size_t ref_scan_hits(const uint8_t *ctg, size_t n, uint32_t k,
                     const uint64_t *splitters, size_t n_spl,
                     size_t cap, uint64_t *hit_pos, uint64_t *hit_dir, uint64_t *hit_rc)
{
    hash_set_lp<uint64_t, std::equal_to<uint64_t>, MurMur64Hash> hs(~0ull, 16ull, 0.4, std::equal_to<uint64_t>{}, MurMur64Hash{});
    bloom_set_t bloom;
    for (size_t i = 0; i < n_spl; ++i)
        hs.insert(splitters[i]);
    bloom.resize((uint64_t)(hs.size() / 0.25));
    bloom.insert(hs.begin(), hs.end());

    CKmer kmer(k, kmer_mode_t::canonical);
    size_t m = 0;
    uint64_t pos = 0;
    for (size_t i = 0; i < n; ++i, ++pos) {
        uint8_t x = ctg[i];
        if (x >> 2)
            kmer.Reset();
        else {
            kmer.insert_canonical(x);
            if (kmer.is_full()) {
                uint64_t d = kmer.data_canonical();
                if (bloom.check(d) && hs.check(d)) {
                    if (m < cap) {
                        hit_pos[m] = pos;
                        hit_dir[m] = kmer.data_dir();
                        CKmer t = kmer;
                        t.swap_dir_rc();
                        hit_rc[m] = t.data_dir();
                    }
                    ++m;
                    kmer.Reset();
                }
            }
        }
    }
    return m;
}

} // extern "C"
