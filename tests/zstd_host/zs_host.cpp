// zs_host.cpp -- TEST INFRASTRUCTURE: the zstd level-17 encoder headers of agc_amd/csrc/zstd/ compiled for the HOST, so that
// the -m "not gpu" tests can compare them with the image's libzstd 1.4.9 (sequences: ZSTD_generateSequences, frames:
// ZSTD_compressCCtx) without a GPU.  The product compiles the same headers with hipcc into libagc_hip.so and never loads this.
#include "../../agc_amd/csrc/zstd/zs_opt.h"
#include <stdlib.h>
#include <vector>

using namespace zs;

extern "C" {

// parser only: sequences as (offCode, litLength, matchLength) triples; returns the number of sequences, *last_lits = trailing literals
int zs_host_parse(const uint8_t *src, uint32_t n, const uint32_t *cparams7, uint32_t *out_seq3, uint32_t cap, uint32_t *last_lits)
{
    OptWs w;
    memset(&w, 0, sizeof(w));
    w.cp = {cparams7[0], cparams7[1], cparams7[2], cparams7[3], cparams7[4], cparams7[5], cparams7[6]};
    const U32 hl3 = w.cp.minMatch == 3 ? (HASHLOG3_MAX < w.cp.windowLog ? HASHLOG3_MAX : w.cp.windowLog) : 0;
    std::vector<U32> ht((size_t)1 << w.cp.hashLog, 0), ht3((size_t)1 << hl3, 0), ct((size_t)1 << w.cp.chainLog, 0);
    std::vector<Optimal> opt(OPT_NUM + 2);
    std::vector<Match> mt(OPT_NUM + 2);
    std::vector<U32> freq(256 + 36 + 53 + 32, 0);
    std::vector<Seq> seqs(n / 3 + 16);
    std::vector<BYTE> lits(n + 16);
    w.hashTable = ht.data();
    w.hashTable3 = ht3.data();
    w.chainTable = ct.data();
    w.opt = opt.data();
    w.matches = mt.data();
    w.litFreq = freq.data();
    w.litLengthFreq = w.litFreq + 256;
    w.matchLengthFreq = w.litLengthFreq + 36;
    w.offCodeFreq = w.matchLengthFreq + 53;
    w.seqs = seqs.data();
    w.lits = lits.data();
    U32 rep[3] = {1, 4, 8};
    U32 ll = 0;
    compressBlockBt(w, rep, src, n, &ll, [](OptWs &w_, U32 *rep_, const BYTE *s_, U32 n_, int l_) { return compressBlockOpt(w_, rep_, s_, n_, l_); });
    *last_lits = ll;
    for (U32 i = 0; i < w.nSeq && i < cap; ++i) {
        out_seq3[3 * i] = w.seqs[i].offCode;
        out_seq3[3 * i + 1] = w.seqs[i].litLength;
        out_seq3[3 * i + 2] = w.seqs[i].matchLength;
    }
    return (int)w.nSeq;
}
}

#include "../../agc_amd/csrc/zstd/zs_frame.h"

extern "C" {
// whole frame; dst must hold n + 16 bytes; returns the frame size
uint32_t zs_host_compress2(const uint8_t *src, uint32_t n, const uint32_t *cparams7, uint8_t *dst, int loop_nest)
{
    CParams cp = {cparams7[0], cparams7[1], cparams7[2], cparams7[3], cparams7[4], cparams7[5], cparams7[6]};
    const WsLayout L = wsLayout(cp, n);
    std::vector<BYTE> ws(L.total, 0);
    // loop_nest & 2: the micro-step parser with its tables split as on the device (small frequency tables and the first
    // FAST_MATCHES matches of a request outside the workspace)
    if (loop_nest & 2) {
        std::vector<U32> fast(FAST_FREQ_WORDS, 0xDEADBEEFu); // (fast memory is not zeroed on the device either)
        std::vector<Match> fm(FAST_MATCHES);
        return compressFrame(ws.data(), cp, src, n, dst, false, 0, fast.data(), fm.data());
    }
    return compressFrame(ws.data(), cp, src, n, dst, loop_nest != 0);
}
uint32_t zs_host_compress(const uint8_t *src, uint32_t n, const uint32_t *cparams7, uint8_t *dst)
{
    CParams cp = {cparams7[0], cparams7[1], cparams7[2], cparams7[3], cparams7[4], cparams7[5], cparams7[6]};
    const WsLayout L = wsLayout(cp, n);
    std::vector<BYTE> ws(L.total, 0);
    return compressFrame(ws.data(), cp, src, n, dst);
}
}

extern "C" {
// the arithmetic forms of the format's code tables, for every argument the tables cover
void zs_host_code_tables(uint32_t *ll_bits36, uint32_t *ml_bits53, uint32_t *ll_code64, uint32_t *ml_code128)
{
    for (uint32_t c = 0; c < 36; ++c) ll_bits36[c] = LLbits(c);
    for (uint32_t c = 0; c < 53; ++c) ml_bits53[c] = MLbits(c);
    for (uint32_t l = 0; l < 64; ++l) ll_code64[l] = LLcode(l);
    for (uint32_t m = 0; m < 128; ++m) ml_code128[m] = MLcode(m);
}
}

extern "C" {
// the group parser (zs_opt_grp.h) run the host way: G lane states, every segment of the trip for lane 0..G-1 in turn -- the
// same code the kernel runs with one lane state per GPU lane and the exchange record in LDS.  Inputs the group path does not
// take (grpEligible) go through the one-lane parser, as in the kernel's dispatch.
uint32_t zs_host_compress_grp(const uint8_t *src, uint32_t n, const uint32_t *cparams7, uint8_t *dst, int G)
{
    CParams cp = {cparams7[0], cparams7[1], cparams7[2], cparams7[3], cparams7[4], cparams7[5], cparams7[6]};
    const WsLayout L = wsLayout(cp, n);
    std::vector<BYTE> ws(L.total, 0);
    if (!grpEligible(cp, n))
        return compressFrame(ws.data(), cp, src, n, dst);
    std::vector<GLane> lanes(3);
    std::vector<U32> recs(3 * 2 * GRP_RC, 0xDEADBEEFu);
    memset((void *)lanes.data(), 0xA5, sizeof(GLane) * 3); // (lane state is not zeroed on the device either)
    for (int i = 0; i < 3; ++i) {
        lanes[i].j = (U32)i;
        lanes[i].recs = recs.data() + (size_t)i * 2 * GRP_RC;
    }
    GrpX sh;
    memset(&sh, 0xA5, sizeof(sh));
    switch (G) {
    case 1: return compressFrameGrp<1>(lanes.data(), sh, ws.data(), cp, src, n, dst);
    case 2: return compressFrameGrp<2>(lanes.data(), sh, ws.data(), cp, src, n, dst);
    default: return compressFrameGrp<3>(lanes.data(), sh, ws.data(), cp, src, n, dst);
    }
}
}
