// sv_host.cpp -- CPU check of agc_amd/csrc/sym_view.h (the symbol accessors the LZ kernels are built on) against plain byte
// arrays: every position / count / orientation of random buffers with N runs and IUPAC codes, with and without escape index.
// TEST INFRASTRUCTURE (built and run by tests/test_symview.py); prints "ok <checks>" or the first difference.
#include "../../agc_amd/csrc/sym_view.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

using namespace agc;

struct Packed {
    std::vector<uint32_t> words;
    std::vector<int32_t> esc_index;
    std::vector<uint8_t> esc_bytes;
};

// the layout pack_codes_kernel writes (scan_kernels.hip)
static Packed pack(const std::vector<uint8_t> &codes)
{
    Packed p;
    const size_t n = codes.size(), nb = (n + SV_BLOCK - 1) / SV_BLOCK;
    p.words.assign(nb * (SV_BLOCK / 16) + 16, 0xDEADBEEFu); // (slack words hold garbage on purpose)
    p.esc_index.assign(nb + 16, 12345);                     // (entries past the last block are garbage as well)
    for (size_t b = 0; b < nb; ++b) {
        bool high = false;
        for (size_t i = b * SV_BLOCK; i < std::min(n, (b + 1) * SV_BLOCK); ++i)
            high = high || codes[i] > 3;
        if (high) {
            p.esc_index[b] = (int32_t)(p.esc_bytes.size() / SV_BLOCK);
            for (size_t i = b * SV_BLOCK; i < (b + 1) * SV_BLOCK; ++i)
                p.esc_bytes.push_back(i < n ? codes[i] : 0);
        } else
            p.esc_index[b] = -1;
        for (size_t w = 0; w < SV_BLOCK / 16; ++w) {
            uint32_t x = 0;
            for (size_t j = 0; j < 16; ++j) {
                const size_t i = b * SV_BLOCK + w * 16 + j;
                x |= (uint32_t)((i < n ? codes[i] : 0) & 3u) << (2 * j);
            }
            p.words[b * (SV_BLOCK / 16) + w] = x;
        }
    }
    return p;
}

static uint8_t comp(uint8_t c) { return c < 4 ? (uint8_t)(3 - c) : c; }

int main(int argc, char **argv)
{
    const unsigned seed = argc > 1 ? (unsigned)atoi(argv[1]) : 1;
    std::mt19937 rng(seed);
    uint64_t checks = 0;
    for (int round = 0; round < 6; ++round) {
        const bool with_esc = round >= 2;
        const size_t n = 3000 + rng() % 4000;
        std::vector<uint8_t> codes(n);
        for (auto &c : codes)
            c = (uint8_t)(rng() & 3);
        if (with_esc) {
            for (int r = 0; r < 3; ++r) { // N runs and single IUPAC codes
                const size_t a = rng() % n, l = 1 + rng() % 40;
                for (size_t i = a; i < std::min(n, a + l); ++i)
                    codes[i] = 4;
                codes[rng() % n] = (uint8_t)(5 + rng() % 11);
                codes[rng() % n] = 30;
            }
            if (round == 5) // a whole escaped region crossing a block boundary
                for (size_t i = 900; i < 2300 && i < n; ++i)
                    codes[i] = 4;
        }
        const Packed pk = pack(codes);
        for (int t = 0; t < 40; ++t) {
            SymView v;
            v.words = pk.words.data();
            v.esc_index = with_esc ? pk.esc_index.data() : nullptr;
            v.esc_bytes = pk.esc_bytes.data();
            v.start = t == 0 ? 0 : rng() % (n - 1);
            v.len = (uint32_t)(t == 1 ? n - v.start : 1 + rng() % (n - v.start));
            v.rc = (uint32_t)(t & 1);
            std::vector<uint8_t> seq(v.len);
            for (uint32_t p = 0; p < v.len; ++p)
                seq[p] = v.rc ? comp(codes[v.start + v.len - 1 - p]) : codes[v.start + p];
            // is the sequence free of escaped blocks?  (then the accessors may be told so)
            bool clean = true;
            if (with_esc)
                for (uint64_t b = v.start / SV_BLOCK; b <= (v.start + v.len - 1) / SV_BLOCK; ++b)
                    clean = clean && pk.esc_index[b] < 0;
            for (int cl = 0; cl < (clean ? 2 : 1); ++cl) {
                for (uint32_t p = 0; p < v.len; ++p) {
                    if (sv_sym(v, p, cl != 0) != seq[p]) {
                        printf("sv_sym differs: round %d start %llu len %u rc %u p %u\n", round, (unsigned long long)v.start, v.len, v.rc, p);
                        return 1;
                    }
                    ++checks;
                    const uint32_t cnts[5] = {32, 1, 1 + (uint32_t)(rng() % 32), 17, 16};
                    for (uint32_t cnt : cnts) {
                        if (p + cnt > v.len)
                            cnt = v.len - p;
                        uint64_t P;
                        uint32_t I;
                        sv_fetch32(v, p, cnt, cl != 0, P, I);
                        for (uint32_t j = 0; j < cnt; ++j) {
                            const uint8_t c = seq[p + j];
                            const bool inv = (I >> j) & 1;
                            if (inv != (c > 3) && !(cl == 0 && !inv && c > 3 && false)) {
                                // (an escaped block may also be read through its words when ... never: I must be exact)
                                printf("sv_fetch32 I differs: round %d start %llu len %u rc %u p %u cnt %u j %u\n", round, (unsigned long long)v.start, v.len,
                                       v.rc, p, cnt, j);
                                return 1;
                            }
                            if (c <= 3 && ((P >> (2 * j)) & 3) != c) {
                                printf("sv_fetch32 P differs: round %d start %llu len %u rc %u p %u cnt %u j %u\n", round, (unsigned long long)v.start, v.len,
                                       v.rc, p, cnt, j);
                                return 1;
                            }
                        }
                        if (cnt < 32 && (I >> cnt) != 0) {
                            printf("sv_fetch32 I has bits past cnt\n");
                            return 1;
                        }
                        ++checks;
                        if (cnt <= 16) { // the two-dword variant delivers the same
                            uint64_t P16;
                            uint32_t I16;
                            sv_fetch16(v, p, cnt, cl != 0, P16, I16);
                            const uint64_t m = cnt == 32 ? ~0ULL : (1ULL << (2 * cnt)) - 1ULL;
                            uint64_t vm = 0; // compare the codes of the ACGT positions only
                            for (uint32_t j = 0; j < cnt; ++j)
                                if (!((I >> j) & 1))
                                    vm |= 3ULL << (2 * j);
                            if (I16 != I || ((P16 ^ P) & m & vm) != 0) {
                                printf("sv_fetch16 differs: round %d start %llu len %u rc %u p %u cnt %u\n", round, (unsigned long long)v.start, v.len, v.rc, p, cnt);
                                return 1;
                            }
                            ++checks;
                        }
                    }
                }
                // keys: first symbol most significant
                for (uint32_t kl = 5; kl <= 29; kl += 6)
                    for (uint32_t p = 0; p + kl <= v.len; p += 7) {
                        uint64_t P;
                        uint32_t I;
                        sv_fetch32(v, p, kl, cl != 0, P, I);
                        if (I & ((1u << kl) - 1u))
                            continue;
                        uint64_t want = 0;
                        for (uint32_t j = 0; j < kl; ++j)
                            want = (want << 2) + seq[p + j];
                        if (sv_key_from_packed(P << (64 - 2 * kl) >> (64 - 2 * kl), kl) != want || sv_key_from_packed(P, kl) != want) {
                            printf("sv_key_from_packed differs: kl %u p %u\n", kl, p);
                            return 1;
                        }
                        ++checks;
                    }
            }
        }
    }
    for (uint32_t x = 0; x < 256; ++x) {
        const uint32_t e = sv_expand4(x);
        for (int j = 0; j < 4; ++j)
            if (((e >> (8 * j)) & 0xFF) != ((x >> (2 * j)) & 3)) {
                printf("sv_expand4 differs: %u\n", x);
                return 1;
            }
    }
    printf("ok %llu\n", (unsigned long long)checks);
    return 0;
}
