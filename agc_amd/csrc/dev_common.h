// dev_common.h -- shared device structs and helpers (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sym_view.h"

namespace agc {

constexpr uint32_t WAVE = 64;
constexpr uint32_t HASHING_STEP = 4;     // lz_diff.h:38 (USE_SPARSE_HT)
constexpr uint32_t MAX_NO_TRIES = 64;    // lz_diff.h:32
constexpr uint8_t INVALID_SYMBOL = 31;   // lz_diff.h:33
constexpr uint8_t N_CODE = 4;            // lz_diff.h:34
constexpr uint8_t N_RUN_STARTER = 30;    // lz_diff.h:35
constexpr uint32_t MIN_NRUN_LEN = 4;     // lz_diff.h:36

// One registered group reference in HBM.
// Index entries: short form (ref/4 < 65535, lz_diff.cpp:146): u32 = pos16<<16 | fp16,
// long form: u64 = pos32<<32 | fp32.  pos = ref position / 4 exactly as ht16/ht32
// hold it; fp = fingerprint of the key's hash (filter only; every candidate is
// verified against the reference bytes).  All-ones = empty.
// Key filter of a reference: a blocked Bloom filter over the 2-bit keys in its index (KEY_BLOOM_WORDS x 64 bits; per key one
// word and three bits of it, from a cheap hash of the key -- its own hash, the index keeps MurMur64).  A text position whose
// key is not in the filter cannot start a match.  key_filter_kernel turns that into one bit per text position ("may match")
// for the estimate / cost-vector parses, whose texts are mostly literal runs (the non-matching half of a missing-middle
// segment, wrong one-splitter candidates): the parse then skips from one candidate position to the next instead of probing
// the index table in HBM at every position.
// Two such filters with independent hashes, KEY_BLOOM_HALF words each (2 x 32 KiB): a key passes when both hold it.  One filter
// lets 0.4 % of the foreign keys through for the 15 k keys of a 60 kb reference -- 240 exact steps (a table row from HBM each) in
// the 60 kb of a missing-middle segment that belong to the OTHER reference, more than its matching half costs; two: 0.002 %.
// Round 5: the filters grow with the reference -- KEY_BLOOM_HALF words each up to 16 k keys (a 60 kb reference: what a human
// collection has, and what fits the LDS of key_filter_kernel), then the next power of two of keys / 4 words (>= 16 bits per key): the
// 600 kb references of a diverged bacterial collection filled the fixed-size filters to 82 % and 30 % of the foreign positions
// passed both -- a foreign stretch then costs an exact step every third position, tens of milliseconds per text.
constexpr uint32_t KEY_BLOOM_HALF = 4096;
constexpr uint32_t KEY_BLOOM_SHIFT0 = 20; // word index = 32 - shift bits of the hash: 12 bits for KEY_BLOOM_HALF words
__host__ __device__ inline uint32_t key_bloom_shift(uint32_t ref_size)
{
    const uint32_t keys = ref_size / HASHING_STEP + 1;
    uint32_t sh = KEY_BLOOM_SHIFT0;
    while (sh > 8 && (1u << (32 - sh)) < keys / 4)
        --sh;
    return sh;
}
__host__ __device__ inline uint32_t key_bloom_half_words(uint32_t shift) { return 1u << (32 - shift); }
// The filters as 32-bit words (a half = 2 * key_bloom_half_words u32 words): the word from the top bits of one product, the
// three bit numbers from the low 5 bits of bytes 1-3 of a second one (the shifter reads them where they lie: SDWA) -- 10 vector
// instructions per key instead of the ~25 of the 64-bit form of rounds 4-5 (key_filter_kernel tests one key per text position).
__host__ __device__ inline uint32_t key_bloom_mask(uint32_t ml)
{
    return (1u << ((ml >> 8) & 31)) | (1u << ((ml >> 16) & 31)) | (1u << ((ml >> 24) & 31));
}
__host__ __device__ inline void key_bloom_slot(uint64_t key, uint32_t shift, uint32_t &word, uint32_t &mask)
{
    const uint32_t a = (uint32_t)key, b = (uint32_t)(key >> 32);
    const uint32_t h = (a ^ (b * 0x9E3779B1u)) * 0x85EBCA6Bu;
    word = h >> (shift - 1);
    mask = key_bloom_mask(h * 0xC2B2AE35u);
}
__host__ __device__ inline void key_bloom_slot2(uint64_t key, uint32_t shift, uint32_t &word, uint32_t &mask)
{
    const uint32_t a = (uint32_t)key, b = (uint32_t)(key >> 32);
    const uint32_t h = ((a * 0xCC9E2D51u) ^ (b + 0x7F4A7C15u)) * 0x1B873593u;
    word = 2 * key_bloom_half_words(shift) + (h >> (shift - 1));
    mask = key_bloom_mask(h * 0x27D4EB2Fu);
}

struct RefDesc {
    // the reference's symbols in the 2-bit layout (sym_view.h): symbol i at bits [2 (i & 15), +1] of words[i >> 4], followed by
    // at least REF_TAIL_WORDS words of slack.  A reference holding anything outside ACGT (rare: N runs, IUPAC codes) also has
    // esc_index / esc_bytes: esc_index[b] = b when block b (1024 symbols) holds such a symbol -- its symbols are then read one
    // byte each from esc_bytes + 1024 b -- and -1 otherwise.  esc_index == nullptr: every symbol is ACGT.
    // There is no padding with INVALID_SYMBOL (prepare_gen, lz_diff.cpp:48-53): compares are bounded by ref_size instead.
    const uint32_t *words;
    const int32_t *esc_index;
    const uint8_t *esc_bytes;
    const void *table;    // ht_mask+1 entries
    const unsigned long long *bloom; // two filters of key_bloom_half_words(key_bloom_shift(ref_size)) words each (nullptr: none)
    uint32_t ref_size;
    uint32_t ht_mask;
    uint32_t key_len;
    uint32_t min_match_len;
    uint32_t is_short;
    uint32_t valid;
};
constexpr uint32_t REF_TAIL_WORDS = 4;

// One sequence to parse: a view into a 2-bit packed buffer (the sample), read reverse-complemented when text.rc.
struct SegDesc {
    SymView text;
    const unsigned long long *maybe; // estimate / cost vector: one bit per text position, 0 = certainly a literal (nullptr: none)
    uint64_t out_off;     // encode: byte offset in the scratch output; cost vector: cost_t offset
    uint32_t ref_slot;    // index into the RefDesc array
    uint32_t flags;       // bit0: prefix_costs (cost-vector mode)
    uint32_t idx;         // index in the caller's order
    uint32_t pad;
};

// a sequence to copy out of a packed buffer: as bytes (slice_expand_kernel), as 2-bit words from bit 0 (ref_pack_kernel), or
// to count lags in (lag_counts_kernel)
struct ViewJob {
    SymView src;
    void *dst;
};

struct ScanRange {
    uint64_t ctg_begin;   // absolute offset of the contig's first symbol
    uint64_t ctg_end;     // absolute offset one past its last symbol
    uint64_t begin;       // first position (absolute) whose k-mer END this range reports
    uint64_t end;         // one past the last
};

// A sample resident in HBM in the 2-bit layout: symbol i of the buffer (contigs back to back) sits at bits [2*(i & 15),
// +1] of 32-bit word i >> 4.  Blocks of PACK_BLOCK symbols that contain anything outside ACGT (N runs, IUPAC codes) are
// "escaped": kept verbatim, one byte per symbol, in esc_bytes; esc_index[block] = slot there or -1.  0.25 B per symbol for
// clean sequence, 1.25 B for escaped blocks.
constexpr uint32_t PACK_BLOCK = 1024;
struct PackedView {
    const uint32_t *words;
    const int32_t *esc_index;
    const uint8_t *esc_bytes;
    uint64_t n_symbols;
};

// Filter over the last 16 symbols of the splitters (both strands): 128 KiB of LDS, 3 bits per entry inside one 32-bit word.
// Made for the instruction count of the scan's inner loop (one test per text position, round 6): the word's BYTE address is the
// high product masked (no shift), the three bit numbers are the low 5 bits of bytes 1-3 of the low product (the shifter reads
// them where they lie: v_lshlrev_b32_sdwa) -- 11 vector instructions per position instead of 15.5 (profiles/EXPERIMENTS.md).
constexpr uint32_t SBLOOM_WORDS = 32768;
__host__ __device__ inline void sbloom_addr_mask(uint32_t w16, uint32_t &byte_addr, uint32_t &mask)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t mh = __umulhi(w16, 0x9E3779B1u);
#else
    const uint32_t mh = (uint32_t)(((uint64_t)w16 * 0x9E3779B1u) >> 32);
#endif
    const uint32_t ml = w16 * 0x85EBCA6Bu;
    byte_addr = mh & ((SBLOOM_WORDS - 1) << 2);
    mask = (1u << ((ml >> 8) & 31)) | (1u << ((ml >> 16) & 31)) | (1u << ((ml >> 24) & 31));
}
__host__ __device__ inline void sbloom_slot(uint32_t w16, uint32_t &word, uint32_t &mask)
{
    uint32_t a;
    sbloom_addr_mask(w16, a, mask);
    word = a >> 2;
}

struct ScanHit {
    uint64_t pos;         // absolute offset of the LAST symbol of the k-mer
    uint64_t dir;         // left-aligned, as CKmer::kmer_dir (kmer.h:284-301)
    uint64_t rc;          // left-aligned, as CKmer::kmer_rc
};

// MurMur64Hash, src/common/utils.h:164-176 (bit-relevant: LZ index slots)
__host__ __device__ inline uint64_t murmur64(uint64_t h)
{
    h ^= h >> 33;
    h *= 0xff51afd7ed558ccdULL;
    h ^= h >> 33;
    h *= 0xc4ceb9fe1a85ec53ULL;
    h ^= h >> 33;
    return h;
}

// hash of the build's own splitter table / bloom (free choice: membership is exact)
__host__ __device__ inline uint64_t splitter_hash(uint64_t x)
{
    x ^= x >> 29;
    x *= 0x9E3779B97F4A7C15ULL;
    x ^= x >> 32;
    return x;
}

constexpr uint32_t BLOOM_WORDS = 16384;  // 64 KiB of LDS

// first-level (LDS) filter slot of a canonical k-mer: one 32-bit multiply, no 64-bit arithmetic
// (the scan evaluates it for every base; quality only affects the false-positive rate)
__host__ __device__ inline void bloom_slot(uint32_t can_hi, uint32_t can_lo, uint32_t &word, uint32_t &mask)
{
    const uint32_t m = (can_hi ^ (can_lo >> 15) ^ (can_lo << 7)) * 0x9E3779B1u;
    word = m >> 18;                                   // 14 bits
    mask = (1u << ((m >> 13) & 31)) | (1u << ((m >> 8) & 31));
}

// second-level filter in global memory (1 MiB, stays in L2): cuts the exact-table probes of the
// LDS bloom's false positives by ~200x at 50 k splitters
constexpr uint32_t BLOOM2_WORDS = 262144;

__host__ __device__ inline void bloom2_slot(uint64_t h, uint32_t &word, uint32_t &mask)
{
    word = (uint32_t)(h >> 20) & (BLOOM2_WORDS - 1);
    mask = (1u << ((uint32_t)(h >> 10) & 31)) | (1u << ((uint32_t)(h >> 15) & 31));
}

} // namespace agc
