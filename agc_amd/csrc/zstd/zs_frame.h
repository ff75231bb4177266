// Derived from Zstandard 1.4.9 (Copyright (c) 2016-present, Facebook, Inc.; BSD license): see NOTICE in this directory.
// zs_frame.h -- one zstd frame as ZSTD_compressCCtx(level) of libzstd 1.4.9 writes it for an input of at most one block
// (<= 128 KiB) with the bt* strategies: frame header (zstd_compress.c: ZSTD_writeFrameHeader), the block
// (ZSTD_compressBlock_internal) and its header (ZSTD_compress_frameChunk).  Everything works inside a caller-provided
// workspace; nothing is allocated.
#pragma once
#include "zs_opt.h"
#include "zs_opt_sm.h"
#include "zs_opt_grp.h"
#include "zs_entropy.h"

namespace zs {

// bytes of workspace one frame needs (all tables of OptWs + EntWs + the sequence store), 16-byte aligned pieces
ZHD size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

struct WsLayout {
    size_t hashTable, hashTable3, chainTable, opt, matches, freqs, seqs, lits, codes, ent, bkfw, total;
};

ZHD WsLayout wsLayout(const CParams &cp, U32 srcSize)
{
    WsLayout L;
    const U32 hl3 = cp.minMatch == 3 ? (HASHLOG3_MAX < cp.windowLog ? HASHLOG3_MAX : cp.windowLog) : 0;
    size_t o = 0;
    L.hashTable = o;
    o += align16(((size_t)4) << cp.hashLog);
    L.hashTable3 = o;
    o += align16(((size_t)4) << hl3);
    L.chainTable = o;
    o += align16(((size_t)4) << cp.chainLog);
    L.opt = o;
    o += align16(sizeof(Optimal) * (OPT_NUM + 2));
    L.matches = o;
    o += align16(sizeof(Match) * (OPT_NUM + 2));
    L.freqs = o;
    o += align16(4 * (256 + 36 + 53 + 32));
    L.seqs = o;
    o += align16(sizeof(Seq) * ((size_t)srcSize / 3 + 16));
    L.lits = o;
    o += align16((size_t)srcSize + 32);
    L.codes = o;
    o += align16(3 * ((size_t)srcSize / 3 + 16));
    L.ent = o;
    o += align16(sizeof(EntWs));
    L.bkfw = o; // (a U32 and a U16 per position of a chunk: zs_opt_grp.h)
    o += align16(6 * (size_t)(OPT_NUM + 2) + 16);
    L.total = o;
    return L;
}

// upper bound of a frame: header (<= 9) + block header (3) + raw block
// room a frame's buffer must have: the frame itself never exceeds srcSize + 12 (a raw block), but a Huffman attempt that turns out
// longer than the literals is only cut short at the literals' size (zs_entropy.h: hufCompress1X) behind up to 17 bytes of headers
ZHD U32 frameBound(U32 srcSize) { return srcSize + 48; }

// frame header: magic, descriptor, [window], content size (contentSizeFlag = 1, no checksum, no dictID); returns the end
ZFN BYTE *writeFrameHeader(BYTE *op, const CParams &cp, U32 srcSize)
{
    const U32 windowSize = 1u << cp.windowLog;
    const U32 singleSegment = windowSize >= srcSize;
    const U32 fcsCode = (srcSize >= 256) + (srcSize >= 65536 + 256);
    op[0] = 0x28;
    op[1] = 0xB5;
    op[2] = 0x2F;
    op[3] = 0xFD;
    op[4] = (BYTE)((singleSegment << 5) + (fcsCode << 6));
    op += 5;
    if (!singleSegment)
        *op++ = (BYTE)((cp.windowLog - 10) << 3);
    switch (fcsCode) {
    case 0:
        if (singleSegment)
            *op++ = (BYTE)srcSize;
        break;
    case 1:
        op[0] = (BYTE)(srcSize - 256);
        op[1] = (BYTE)((srcSize - 256) >> 8);
        op += 2;
        break;
    default:
        op[0] = (BYTE)srcSize;
        op[1] = (BYTE)(srcSize >> 8);
        op[2] = (BYTE)(srcSize >> 16);
        op[3] = (BYTE)(srcSize >> 24);
        op += 4;
        break;
    }
    return op;
}

// the block header and, when the block did not shrink, the raw block (ZSTD_noCompressBlock); returns the frame's size
ZFN U32 finishFrame(BYTE *dst, BYTE *op, U32 cSize, const BYTE *src, U32 srcSize)
{
    if (cSize == 0) {
        const U32 h = 1 + (0u << 1) + (srcSize << 3);
        op[0] = (BYTE)h;
        op[1] = (BYTE)(h >> 8);
        op[2] = (BYTE)(h >> 16);
        for (U32 i = 0; i < srcSize; ++i)
            op[3 + i] = src[i];
        return (U32)(op + 3 + srcSize - dst);
    }
    const U32 h = 1 + (2u << 1) + (cSize << 3);
    op[0] = (BYTE)h;
    op[1] = (BYTE)(h >> 8);
    op[2] = (BYTE)(h >> 16);
    return (U32)(op + 3 + cSize - dst);
}

// the level's frame for src[0..srcSize), 7 <= ... any srcSize <= BLOCKSIZE_MAX, written to dst (frameBound bytes).
// ws: wsLayout(cp, srcSize).total bytes, hashTable / hashTable3 / chainTable regions ZEROED by the caller.
// loop_nest: parse with the plain loop nest (zs_opt.h) instead of the micro-step loop (zs_opt_sm.h); same bytes
// fast_freqs: FAST_FREQ_WORDS words of fast memory for the three small statistics tables the price loops read all the time
// (lit-length, match-length, offset-code frequencies) -- the kernel passes a slice of LDS; nullptr = they live in ws
constexpr U32 FAST_FREQ_WORDS = 36 + 53 + 32;
constexpr U32 FAST_MATCHES = 16;                                  // matches of a request kept in fast memory (FASTM)
constexpr U32 FAST_WORDS = FAST_FREQ_WORDS + 2 * FAST_MATCHES;    // per frame, when both are used (odd: lanes on different banks)
// FAST (device only): the tables are at fast_freqs; !FAST: in ws (a launch that must leave the LDS to kernels of other streams)
// FASTM: the first FAST_MATCHES matches of a request are kept at fast_freqs + FAST_FREQ_WORDS (host: fast_matches, a test hook)
template <bool FAST = true, bool FASTM = false>
ZFN U32 compressFrame(BYTE *ws, const CParams &cp, const BYTE *src, U32 srcSize, BYTE *dst, bool loop_nest = false, U32 debug = 0,
                      U32 *fast_freqs = nullptr, Match *fast_matches = nullptr)
{
    BYTE *op = writeFrameHeader(dst, cp, srcSize);
    if (srcSize == 0) { // ZSTD_writeEpilogue: one empty raw block marked last
        op[0] = 1;
        op[1] = 0;
        op[2] = 0;
        return (U32)(op + 3 - dst);
    }
    U32 cSize = 0;
    if (srcSize >= 7) { // MIN_CBLOCK_SIZE + ZSTD_blockHeaderSize + 1: smaller blocks are not even tried
        const WsLayout L = wsLayout(cp, srcSize);
        OptWs w;
        w.hashTable = (U32 *)(ws + L.hashTable);
        w.hashTable3 = (U32 *)(ws + L.hashTable3);
        w.chainTable = (U32 *)(ws + L.chainTable);
        w.opt = (Optimal *)(ws + L.opt);
        w.matches = (Match *)(ws + L.matches);
        w.litFreq = (U32 *)(ws + L.freqs);
#if defined(__HIP_DEVICE_COMPILE__)
        // (decided at compile time on the device so that the compiler sees ONE address space behind each pointer)
        if (FAST) {
            w.litLengthFreq = fast_freqs;
            w.matchLengthFreq = fast_freqs + 36;
            w.offCodeFreq = fast_freqs + 36 + 53;
        } else {
            w.litLengthFreq = w.litFreq + 256;
            w.matchLengthFreq = w.litLengthFreq + 36;
            w.offCodeFreq = w.matchLengthFreq + 53;
        }
#else
        w.litLengthFreq = fast_freqs ? fast_freqs : w.litFreq + 256;
        w.matchLengthFreq = w.litLengthFreq + 36;
        w.offCodeFreq = w.matchLengthFreq + 53;
#endif
#if defined(__HIP_DEVICE_COMPILE__)
        w.fastMatches = FASTM ? (Match *)(fast_freqs + FAST_FREQ_WORDS) : nullptr;
        w.fastMatchCap = FASTM ? FAST_MATCHES : 0;
#else
        w.fastMatches = fast_matches;
        w.fastMatchCap = fast_matches ? FAST_MATCHES : 0;
#endif
        w.seqs = (Seq *)(ws + L.seqs);
        w.lits = ws + L.lits;
        w.cp = cp;
        U32 rep[3] = {1, 4, 8};
        U32 lastLits = 0;
        if (loop_nest)
            compressBlockBt(w, rep, src, srcSize, &lastLits, [](OptWs &w_, U32 *rep_, const BYTE *s_, U32 n_, int l_) { return compressBlockOpt(w_, rep_, s_, n_, l_); });
        else
            compressBlockBt(w, rep, src, srcSize, &lastLits, [](OptWs &w_, U32 *rep_, const BYTE *s_, U32 n_, int l_) { return compressBlockOptSM(w_, rep_, s_, n_, l_); });
        for (U32 i = 0; i < lastLits; ++i) // ZSTD_storeLastLiterals
            w.lits[w.nLits + i] = src[srcSize - lastLits + i];
        w.nLits += lastLits;
        EntWs &e = *(EntWs *)(ws + L.ent);
        if (!(debug & 1)) // (debug bit 0: profiling aid -- parse only, emit a raw block)
        cSize = entropyCompressBlock(e, cp, w.seqs, w.nSeq, w.lits, w.nLits, ws + L.codes, op + 3, srcSize);
    }
    return finishFrame(dst, op, cSize, src, srcSize);
}

// The same frame by a GROUP of G lanes (zs_opt_grp.h): every lane of the group calls this with its own lane state; the leader
// (lanes[0].j == 0 on the device; lanes[0] on the host) writes the frame and returns its size, the others return 0.
// fast_freqs: the group's FAST_FREQ_WORDS words of fast memory (device: LDS) or nullptr (host: the tables live in ws).
template <int G> ZFN U32 compressFrameGrp(GLane *lanes, GrpX &sh, BYTE *ws, const CParams &cp, const BYTE *src, U32 srcSize, BYTE *dst, U32 debug = 0,
                                          U32 *fast_freqs = nullptr)
{
    const bool leader = lanes[0].j == 0;
    BYTE *op = dst;
    if (leader)
        op = writeFrameHeader(dst, cp, srcSize);
    if (srcSize == 0) { // ZSTD_writeEpilogue: one empty raw block marked last
        if (!leader)
            return 0;
        op[0] = 1;
        op[1] = 0;
        op[2] = 0;
        return (U32)(op + 3 - dst);
    }
    U32 cSize = 0;
    if (srcSize >= 8) { // (7-byte inputs -- parsed by the library, never compressible -- take the raw block below as well)
        const WsLayout L = wsLayout(cp, srcSize);
        ZS_GRP_EACH(l)
        OptWs &w = l.w;
        w.hashTable = (U32 *)(ws + L.hashTable);
        w.hashTable3 = (U32 *)(ws + L.hashTable3);
        w.chainTable = (U32 *)(ws + L.chainTable);
        w.opt = (Optimal *)(ws + L.opt);
        w.matches = (Match *)(ws + L.matches);
        w.bk = (U32 *)(ws + L.bkfw);
        w.fw = (U16 *)(w.bk + (OPT_NUM + 2));
        w.litFreq = (U32 *)(ws + L.freqs);
#if defined(__HIP_DEVICE_COMPILE__)
        w.litLengthFreq = fast_freqs; // (always LDS on the device: ONE address space behind the pointer)
#else
        w.litLengthFreq = fast_freqs ? fast_freqs : w.litFreq + 256;
#endif
        w.matchLengthFreq = w.litLengthFreq + 36;
        w.offCodeFreq = w.matchLengthFreq + 53;
        w.fastMatches = nullptr;
        w.fastMatchCap = 0;
        w.seqs = (Seq *)(ws + L.seqs);
        w.lits = ws + L.lits;
        w.cp = cp;
        w.litSum = w.litLengthSum = w.matchLengthSum = w.offCodeSum = 0;
        w.nSeq = 0;
        w.nLits = 0;
        w.idx0 = 1;
        w.dictLimit = 1;
        w.nextToUpdate = 1;
        w.hashLog3 = cp.minMatch == 3 ? (HASHLOG3_MAX < cp.windowLog ? HASHLOG3_MAX : cp.windowLog) : 0;
        ZS_GRP_END
        // btultra2 (ZSTD_initStats_ultra): a first pass over the block collects statistics, its sequences are dropped.  ONE call
        // site in a loop: the parser is inlined once (the loop body is ~100 KB of code as it is)
        const int optLevel = 2; // (grpEligible: btultra / btultra2 only -- the caller sends everything else to compressFrame)
        const U32 passes = (cp.strategy == STRAT_BTULTRA2 && srcSize > PREDEF_THRESHOLD) ? 2 : 1;
        U32 lastLits = 0;
        for (U32 pass = 0; pass < passes; ++pass) {
            U32 rep[3] = {1, 4, 8};
            lastLits = compressBlockOptGrp<G>(lanes, sh, rep, src, srcSize, optLevel);
            if (pass + 1 < passes) {
                ZS_GRP_EACH(l)
                l.w.nSeq = 0;
                l.w.nLits = 0;
                l.w.idx0 += srcSize; // window.base -= srcSize
                l.w.dictLimit += srcSize;
                l.w.nextToUpdate = l.w.dictLimit;
                if (l.j == 0)
                    upscaleStats(l.w);
                ZS_GRP_END
            }
        }
        if (!leader)
            return 0;
        OptWs &w = lanes[0].w;
        for (U32 i = 0; i < lastLits; ++i) // ZSTD_storeLastLiterals
            w.lits[w.nLits + i] = src[srcSize - lastLits + i];
        w.nLits += lastLits;
        EntWs &e = *(EntWs *)(ws + L.ent);
        if (!(debug & 1)) // (debug bit 0: profiling aid -- parse only, emit a raw block)
            cSize = entropyCompressBlock(e, cp, w.seqs, w.nSeq, w.lits, w.nLits, ws + L.codes, op + 3, srcSize);
    }
    if (!leader)
        return 0;
    return finishFrame(dst, op, cSize, src, srcSize);
}

} // namespace zs
