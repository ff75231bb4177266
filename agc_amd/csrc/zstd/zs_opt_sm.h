// Derived from Zstandard 1.4.9 (Copyright (c) 2016-present, Facebook, Inc.; BSD license): see NOTICE in this directory.
// zs_opt_sm.h -- ZSTD_compressBlock_opt_generic (zs_opt.h: compressBlockOpt) rewritten as ONE loop over micro-steps.
//
// Why: on the GPU every lane of a wave compresses its own frame.  The nested loops of the parser (positions of a chunk > tree
// levels > compared bytes; matches > lengths) have data-dependent trip counts, so in the loop nest the lanes of a wave wait
// for each other at every level -- measured: ~5 of 64 lanes active on average, 20x the instructions of one frame.  Here the
// whole parse is a state machine; one trip of the single loop advances every lane by one micro-step of whatever it is doing
// (one tree level, a few price updates, one stored sequence ...), so lanes never wait at inner loop exits and re-converge
// at the loop head.  Frames of equal size need (almost) the same number of micro-steps.
//
// The decisions, their order and every side effect are those of compressBlockOpt (zs_opt.h), which stays as the plain
// restatement of the library's loop nest; tests/test_zstd_frames.py runs both against libzstd.
#pragma once
#include "zs_opt.h"

namespace zs {

#ifndef ZS_SM_PRICE_STEPS
#define ZS_SM_PRICE_STEPS 8
#endif
#ifndef ZS_SM_WALK_LEVELS
#define ZS_SM_WALK_LEVELS 6
#endif
constexpr U32 SM_PRICE_STEPS = ZS_SM_PRICE_STEPS;     // price updates per trip
constexpr U32 SM_WALK_LEVELS = ZS_SM_WALK_LEVELS;     // tree levels per trip
constexpr U32 SM_STORE_SEQS = 3;      // stored sequences per trip
constexpr U32 SM_NOPTR = 0xFFFFFFFFu; // "dummy32": the tree pointer that is no longer written

enum {
    ST_FIND_FIRST = 0, // outer loop head: look for a match at ip
    ST_CUR_BEGIN,      // head of one `cur` iteration of the forward pass
    ST_GETM_BEGIN,     // ZSTD_BtGetAllMatches prologue
    ST_UPD_BEGIN,      // ZSTD_updateTree: next skipped position
    ST_UPD_WALK,       // ZSTD_insertBt1: one tree level
    ST_GETM_REP,       // repcodes, hash3, walk set-up
    ST_WALK,           // ZSTD_insertBtAndGetAllMatches: one tree level
    ST_AFTER_MATCHES,  // back in the parser with nbMatches
    ST_PRICE_FIRST,    // prices of the first matches (position 0 of a chunk)
    ST_PRICE_CUR,      // prices of the matches found at cur
    ST_CUR_NEXT,       // ++cur, loop test
    ST_CHUNK_END,      // _shortestPath: repcodes, backward traversal
    ST_STORE,          // one sequence of the chunk
    ST_DONE
};

// price / off / mlen / litlen of an entry in one 16-byte store (the repcodes of the entry are written later)
ZFN void storeHead(Optimal &dst, const Optimal &o)
{
    struct alignas(16) Head {
        int price;
        U32 off, mlen, litlen;
    };
    Head h = {o.price, o.off, o.mlen, o.litlen};
    *(Head *)&dst = h;
}

ZFN U32 compressBlockOptSM(OptWs &w, U32 rep[3], const BYTE *src, U32 srcSize, int optLevel)
{
    const BYTE *const iend = src + srcSize;
    const U32 ilimit_off = srcSize - 8; // ilimit = iend - 8 (srcSize >= 8 guaranteed by the caller path; smaller blocks never get here)
    const CParams cp = w.cp;
    const U32 sufficient_len = cp.targetLength < OPT_NUM - 1 ? cp.targetLength : OPT_NUM - 1;
    const U32 minMatch = (cp.minMatch == 3) ? 3 : 4;
    const U32 mls = cp.minMatch <= 3 ? 3 : (cp.minMatch == 4 ? 4 : (cp.minMatch == 5 ? 5 : 6));
    const U32 btMask = (1u << (cp.chainLog - 1)) - 1;
    U32 *const bt = w.chainTable;
    Optimal *const opt = w.opt;
    Match *const matches = w.matches;
    Match *const fm = w.fastMatches; // the first fmCap matches of a request (a position rarely has more than a handful)
    const U32 fmCap = w.fastMatchCap;
#define ZS_M_SET(i_, off_, len_)                 \
    do {                                         \
        const U32 mi_ = (i_);                    \
        if (mi_ < fmCap) {                       \
            fm[mi_].off = (off_);                \
            fm[mi_].len = (len_);                \
        } else {                                 \
            matches[mi_].off = (off_);           \
            matches[mi_].len = (len_);           \
        }                                        \
    } while (0)
#define ZS_M_GET(i_, off_, len_)                 \
    do {                                         \
        const U32 mi_ = (i_);                    \
        if (mi_ < fmCap) {                       \
            (off_) = fm[mi_].off;                \
            (len_) = fm[mi_].len;                \
        } else {                                 \
            (off_) = matches[mi_].off;           \
            (len_) = matches[mi_].len;           \
        }                                        \
    } while (0)
    U32 nextToUpdate3 = w.nextToUpdate;

    // parser state
    U32 ip = 0, anchor = 0;       // offsets in src
    U32 cur = 0, last_pos = 0;
    Optimal lastSequence;
    lastSequence.price = 0;
    lastSequence.off = lastSequence.mlen = lastSequence.litlen = 0;
    lastSequence.rep[0] = lastSequence.rep[1] = lastSequence.rep[2] = 0;
    bool inChunk = false;         // the match request comes from the forward pass (cur) rather than from the outer loop
    // match request
    U32 q_current = 0;            // index of the position
    U32 q_ll0 = 0, q_litlen = 0;
    U32 q_rep[3] = {0, 0, 0};
    U32 basePrice = 0;
    U32 nbMatches = 0;
    // tree walk
    U32 wk_current = 0, matchIndex = 0, clSmaller = 0, clLarger = 0, smallerPtr = 0, largerPtr = 0, matchEndIdx = 0, bestLength = 0, nbCompares = 0,
        btLow = 0, lowLimit = 0, mnum = 0, upd_idx = 0;
    // price loops (the match being priced is cached: pm_*)
    U32 pr_matchNb = 0, pr_pos = 0, pr_literalsPrice = 0, pm_off = 0, pm_len = 0, pm_start = 0, cur_litlen_back = 0;
    U32 last_m_off = 0, last_m_len = 0; // the longest match of the current request (matches[nbMatches - 1])
    // store loop
    U32 storePos = 0, storeEnd = 0;
    // the 8 source bytes at the position being worked on, read at the head of the position (together with the price-table
    // entries) so that the hashes, the repcode tests and the literal price of the NEXT position need no load of their own
    U64 p8 = 0;
    U32 p8_pos = 0xFFFFFFFFu; // offset in src p8 was read at

    rescaleFreqs(w, src, srcSize, optLevel);
    ip += (w.idx0 == w.dictLimit);

    U32 state = ST_FIND_FIRST;
    if (srcSize < 8)
        state = ST_DONE; // ip < ilimit never holds
    while (state != ST_DONE) {
        // the compiler must not thread "next state = X" into a jump to X: that would rebuild the loop nest (a lane spinning in
        // its tree walk while the others wait).  An opaque copy of the state keeps the dispatch at the loop head.
#ifdef __HIPCC__
        asm volatile("" : "+v"(state));
#else
        asm volatile("" : "+r"(state));
#endif
        // One trip = the blocks below in the order a position flows through them, each one guarded by the state it serves: a
        // lane on the common path (next position -> literal price -> repcodes / hash3 -> a few tree levels -> prices) finishes a
        // whole position per trip, a lane in a long tree walk, a long price run or a chunk end uses further trips for it.
        for (U32 sq_ = 0; sq_ < SM_STORE_SEQS && state == ST_STORE; ++sq_) do { // a few sequences of the finished chunk per trip
            if (storePos > storeEnd) {
                setBasePrices(w, optLevel);
                state = ST_FIND_FIRST;
                break;
            }
            const U32 llen = opt[storePos].litlen;
            const U32 mlen = opt[storePos].mlen;
            const U32 offCode = opt[storePos].off;
            if (mlen == 0) { // only literals => must be last "sequence", actually starting a new stream of sequences
                ip = anchor + llen;
            } else {
                updateStats(w, llen, src + anchor, offCode, mlen);
                storeSeq(w, llen, src + anchor, offCode, mlen);
                anchor += llen + mlen;
                ip = anchor;
            }
            storePos++;
        } while (0);
        if (state == ST_CUR_NEXT) do {
            cur++;
            if (cur <= last_pos)
                state = ST_CUR_BEGIN;
            else { // the forward loop ran out
                lastSequence = opt[last_pos];
                const U32 tl = lastSequence.litlen + lastSequence.mlen;
                cur = last_pos > tl ? last_pos - tl : 0; // single sequence, and it starts before `ip`
                state = ST_CHUNK_END;
            }
        } while (0);
        if (state == ST_FIND_FIRST) do {
            if (!(srcSize >= 8 && ip < ilimit_off)) {
                state = ST_DONE;
                break;
            }
            p8 = read64(src + ip); // (ip < srcSize - 8)
            p8_pos = ip;
            q_litlen = ip - anchor;
            q_ll0 = !q_litlen;
            q_current = ip + w.idx0;
            q_rep[0] = rep[0];
            q_rep[1] = rep[1];
            q_rep[2] = rep[2];
            inChunk = false;
            state = ST_GETM_BEGIN;
        } while (0);
        if (state == ST_CUR_BEGIN) do {
            const U32 inr = ip + cur;
            // everything this position reads from memory is asked for up front: the literal before it (usually the first byte
            // of the previous position's window) and its frequency, the 8 bytes at the position, the two price-table entries
            const U32 lit_byte = (p8_pos == inr - 1) ? (U32)(p8 & 0xFF) : (U32)src[inr - 1];
            const U32 lit_freq = (w.priceType == zop_predef) ? 0 : w.litFreq[lit_byte];
            if (inr <= ilimit_off) {
                p8 = read64(src + inr);
                p8_pos = inr;
            }
            // the two entries are read ONCE, worked on in registers and written back once (every further look at opt[cur] in this
            // block uses the copy: a reload after a store is a round trip to memory on this hardware)
            const Optimal op = opt[cur - 1];
            Optimal oc = opt[cur];
            {
                const U32 litlen = (op.mlen == 0) ? op.litlen + 1 : 1;
                // rawLiteralsCost(src + inr - 1, 1)
                const U32 lit_cost = (w.priceType == zop_predef) ? 6 * BITCOST_MULTIPLIER : w.litSumBasePrice - weight(lit_freq, optLevel);
                const int price = op.price + (int)lit_cost + (int)litLengthPrice(litlen, w, optLevel) -
                                  (int)litLengthPrice(litlen - 1, w, optLevel);
                if (price <= oc.price) {
                    oc.mlen = 0;
                    oc.off = 0;
                    oc.litlen = litlen;
                    oc.price = price;
                }
            }
            if (oc.mlen != 0) {
                const U32 prev = cur - oc.mlen;
                U32 pr[3];
                pr[0] = opt[prev].rep[0];
                pr[1] = opt[prev].rep[1];
                pr[2] = opt[prev].rep[2];
                updateRep(oc.rep, pr, oc.off, oc.litlen == 0);
            } else {
                for (U32 i = 0; i < REP_NUM; ++i)
                    oc.rep[i] = op.rep[i];
            }
            opt[cur] = oc;
            if (inr > ilimit_off) { // last match must start at a minimum distance of 8 from oend
                state = ST_CUR_NEXT;
                break;
            }
            if (cur == last_pos) { // `break` of the forward loop
                lastSequence = oc;
                const U32 tl = lastSequence.litlen + lastSequence.mlen;
                cur = last_pos > tl ? last_pos - tl : 0;
                state = ST_CHUNK_END;
                break;
            }
            if ((optLevel == 0) && (opt[cur + 1].price <= oc.price + (int)(BITCOST_MULTIPLIER / 2))) {
                state = ST_CUR_NEXT; // skip unpromising positions
                break;
            }
            q_ll0 = (oc.mlen != 0);
            q_litlen = (oc.mlen == 0) ? oc.litlen : 0;
            cur_litlen_back = (oc.mlen == 0) ? oc.litlen : 0; // what `cur -= ...` of the early chunk end needs
            basePrice = (U32)oc.price + litLengthPrice(0, w, optLevel);
            q_current = inr + w.idx0;
            q_rep[0] = oc.rep[0];
            q_rep[1] = oc.rep[1];
            q_rep[2] = oc.rep[2];
            inChunk = true;
            state = ST_GETM_BEGIN;
        } while (0);
        if (state == ST_GETM_BEGIN) do {
            if (q_current < w.nextToUpdate) { // skipped area
                nbMatches = 0;
                state = ST_AFTER_MATCHES;
                break;
            }
            upd_idx = w.nextToUpdate;
            state = ST_UPD_BEGIN;
        } while (0);
        for (U32 it_ = 0; it_ < 2 && (state == ST_UPD_BEGIN || state == ST_UPD_WALK); ++it_) { // skipped positions: ZSTD_updateTree
            if (state == ST_UPD_BEGIN) do {
                if (!(upd_idx < q_current)) {
                    w.nextToUpdate = q_current;
                    state = ST_GETM_REP;
                    break;
                }
                // ZSTD_insertBt1 set-up for position upd_idx
                wk_current = upd_idx;
                const BYTE *const p = src + (wk_current - w.idx0);
                const U32 h = hashPtr(p, cp.hashLog, mls);
                matchIndex = w.hashTable[h];
                clSmaller = clLarger = 0;
                btLow = btMask >= wk_current ? 0 : wk_current - btMask;
                smallerPtr = 2 * (wk_current & btMask);
                largerPtr = smallerPtr + 1;
                lowLimit = w.dictLimit;
                matchEndIdx = wk_current + 8 + 1;
                bestLength = 8;
                nbCompares = 1u << cp.searchLog;
                w.hashTable[h] = wk_current;
                state = ST_UPD_WALK;
            } while (0);
            if (state == ST_UPD_WALK) do {
                bool ended = true;
                if (nbCompares && (matchIndex >= lowLimit)) {
                    nbCompares--;
                    const BYTE *const p = src + (wk_current - w.idx0);
                    const U32 nextPtr = 2 * (matchIndex & btMask);
                    U32 matchLength = clSmaller < clLarger ? clSmaller : clLarger;
                    const BYTE *const match = src + (matchIndex - w.idx0);
                    // both children are read before the stores below (a store to this walk's own pointers never hits the node read here)
                U32 childSmaller, childLarger;
                {
                    const U64 pair = *(const U64 *)(bt + nextPtr);
                    childLarger = (U32)pair;          // nextPtr[0]
                    childSmaller = (U32)(pair >> 32); // nextPtr[1]
                }
                U32 pByte = 0, mByte = 0;
                bool differ = false;
                matchLength += countEx(p + matchLength, match + matchLength, iend, &pByte, &mByte, &differ);
                    if (matchLength > bestLength) {
                        bestLength = matchLength;
                        if (matchLength > matchEndIdx - matchIndex)
                            matchEndIdx = matchIndex + matchLength;
                    }
                    if (p + matchLength != iend) {
                        ended = false;
                        if (mByte < pByte) { // match[matchLength] < p[matchLength] (differ holds: p + matchLength != iend)
                            if (smallerPtr != SM_NOPTR)
                                bt[smallerPtr] = matchIndex;
                            clSmaller = matchLength;
                            if (matchIndex <= btLow) {
                                smallerPtr = SM_NOPTR;
                                ended = true;
                            } else {
                                smallerPtr = nextPtr + 1;
                                matchIndex = childSmaller;
                            }
                        } else {
                            if (largerPtr != SM_NOPTR)
                                bt[largerPtr] = matchIndex;
                            clLarger = matchLength;
                            if (matchIndex <= btLow) {
                                largerPtr = SM_NOPTR;
                                ended = true;
                            } else {
                                largerPtr = nextPtr;
                                matchIndex = childLarger;
                            }
                        }
                    }
                }
                if (ended) {
                    if (smallerPtr != SM_NOPTR)
                        bt[smallerPtr] = 0;
                    if (largerPtr != SM_NOPTR)
                        bt[largerPtr] = 0;
                    U32 positions = 0;
                    if (bestLength > 384)
                        positions = bestLength - 384 < 192 ? bestLength - 384 : 192;
                    const U32 adv = matchEndIdx - (wk_current + 8);
                    upd_idx += positions > adv ? positions : adv;
                    state = ST_UPD_BEGIN;
                }
            } while (0);
        }
        if (state == ST_GETM_REP) do {
            // ZSTD_insertBtAndGetAllMatches up to the tree walk
            wk_current = q_current;
            const BYTE *const p = src + (wk_current - w.idx0);
            const U64 pv = (p8_pos == wk_current - w.idx0) ? p8 : read64(p); // (p <= iend - 8)
            const U32 h = mls == 5 ? hash5(pv, cp.hashLog) : mls == 6 ? hash6(pv, cp.hashLog) : hash4((U32)pv, cp.hashLog);
            matchIndex = w.hashTable[h];
            clSmaller = clLarger = 0;
            const U32 dictLimit = w.dictLimit;
            btLow = (btMask >= wk_current) ? 0 : wk_current - btMask;
            const U32 maxDistance = 1u << cp.windowLog;
            const U32 windowLow = (wk_current - dictLimit > maxDistance) ? wk_current - maxDistance : dictLimit;
            lowLimit = windowLow ? windowLow : 1; // matchLow
            smallerPtr = 2 * (wk_current & btMask);
            largerPtr = smallerPtr + 1;
            matchEndIdx = wk_current + 8 + 1;
            mnum = 0;
            nbCompares = 1u << cp.searchLog;
            bestLength = minMatch - 1; // lengthToBeat - 1
            bool done = false;
            {
                const U32 lastR = REP_NUM + q_ll0;
                for (U32 repCode = q_ll0; repCode < lastR; repCode++) {
                    const U32 repOffset = (repCode == REP_NUM) ? (q_rep[0] - 1) : q_rep[repCode];
                    const U32 repIndex = wk_current - repOffset;
                    U32 repLen = 0;
                    if (repOffset - 1 /* intentional overflow, discards 0 and -1 */ < wk_current - dictLimit) {
                        if ((repIndex >= windowLow) & ((minMatch == 3 ? ((U32)pv << 8) : (U32)pv) == readMINMATCH(p - repOffset, minMatch)))
                            repLen = count(p + minMatch, p + minMatch - repOffset, iend) + minMatch;
                    }
                    if (repLen > bestLength) {
                        bestLength = repLen;
                        ZS_M_SET(mnum, repCode - q_ll0, repLen);
                        last_m_off = repCode - q_ll0;
                        last_m_len = repLen;
                        mnum++;
                        if ((repLen > sufficient_len) | (p + repLen == iend)) {
                            done = true; // best possible
                            break;
                        }
                    }
                }
            }
            if (!done && (mls == 3) && (bestLength < mls)) { // HC3 match finder
                // ZSTD_insertAndFindFirstIndexHash3 (the hash of the position itself from the window)
                U32 matchIndex3;
                {
                    const U32 h3 = hash3((U32)pv, w.hashLog3);
                    for (U32 idx = nextToUpdate3; idx < wk_current; ++idx)
                        w.hashTable3[hash3(read32(src + (idx - w.idx0)), w.hashLog3)] = idx;
                    nextToUpdate3 = wk_current;
                    matchIndex3 = w.hashTable3[h3];
                }
                if ((matchIndex3 >= lowLimit) & (wk_current - matchIndex3 < (1u << 18))) {
                    const BYTE *const match = src + (matchIndex3 - w.idx0);
                    const U32 mlen = count(p, match, iend);
                    if (mlen >= mls) {
                        bestLength = mlen;
                        ZS_M_SET(0, (wk_current - matchIndex3) + REP_MOVE, mlen);
                        last_m_off = (wk_current - matchIndex3) + REP_MOVE;
                        last_m_len = mlen;
                        mnum = 1;
                        if ((mlen > sufficient_len) | (p + mlen == iend)) {
                            w.nextToUpdate = wk_current + 1; // skip insertion
                            done = true;
                        }
                    }
                }
            }
            if (done) {
                nbMatches = mnum;
                state = ST_AFTER_MATCHES;
                break;
            }
            w.hashTable[h] = wk_current;
            state = ST_WALK;
        } while (0);
        for (U32 lv_ = 0; lv_ < SM_WALK_LEVELS && state == ST_WALK; ++lv_) { // a few levels per trip
            bool ended = true;
            if (nbCompares && (matchIndex >= lowLimit)) {
                nbCompares--;
                const BYTE *const p = src + (wk_current - w.idx0);
                const U32 nextPtr = 2 * (matchIndex & btMask);
                U32 matchLength = clSmaller < clLarger ? clSmaller : clLarger;
                const BYTE *const match = src + (matchIndex - w.idx0);
                // both children are read before the stores below (a store to this walk's own pointers never hits the node read here)
                U32 childSmaller, childLarger;
                {
                    const U64 pair = *(const U64 *)(bt + nextPtr);
                    childLarger = (U32)pair;          // nextPtr[0]
                    childSmaller = (U32)(pair >> 32); // nextPtr[1]
                }
                U32 pByte = 0, mByte = 0;
                bool differ = false;
                matchLength += countEx(p + matchLength, match + matchLength, iend, &pByte, &mByte, &differ);
                bool brk = false;
                if (matchLength > bestLength) {
                    if (matchLength > matchEndIdx - matchIndex)
                        matchEndIdx = matchIndex + matchLength;
                    bestLength = matchLength;
                    ZS_M_SET(mnum, (wk_current - matchIndex) + REP_MOVE, matchLength);
                    last_m_off = (wk_current - matchIndex) + REP_MOVE;
                    last_m_len = matchLength;
                    mnum++;
                    if ((matchLength > OPT_NUM) | (p + matchLength == iend))
                        brk = true; // drop, to preserve bt consistency
                }
                if (!brk) {
                    ended = false;
                    if (mByte < pByte) { // match[matchLength] < p[matchLength] (differ holds: p + matchLength != iend)
                        if (smallerPtr != SM_NOPTR)
                            bt[smallerPtr] = matchIndex;
                        clSmaller = matchLength;
                        if (matchIndex <= btLow) {
                            smallerPtr = SM_NOPTR;
                            ended = true;
                        } else {
                            smallerPtr = nextPtr + 1;
                            matchIndex = childSmaller;
                        }
                    } else {
                        if (largerPtr != SM_NOPTR)
                            bt[largerPtr] = matchIndex;
                        clLarger = matchLength;
                        if (matchIndex <= btLow) {
                            largerPtr = SM_NOPTR;
                            ended = true;
                        } else {
                            largerPtr = nextPtr;
                            matchIndex = childLarger;
                        }
                    }
                }
            }
            if (ended) {
                if (smallerPtr != SM_NOPTR)
                    bt[smallerPtr] = 0;
                if (largerPtr != SM_NOPTR)
                    bt[largerPtr] = 0;
                w.nextToUpdate = matchEndIdx - 8; // skip repetitive patterns
                nbMatches = mnum;
                state = ST_AFTER_MATCHES;
            }
        }
        if (state == ST_AFTER_MATCHES) do {
            if (!inChunk) {
                if (!nbMatches) {
                    ip++;
                    state = ST_FIND_FIRST;
                    break;
                }
                for (U32 i = 0; i < REP_NUM; i++)
                    opt[0].rep[i] = rep[i];
                opt[0].mlen = 0;
                opt[0].litlen = q_litlen;
                opt[0].price = (int)litLengthPrice(q_litlen, w, optLevel);
                const U32 maxML = last_m_len;
                const U32 maxOffset = last_m_off;
                if (maxML > sufficient_len) { // large match -> immediate encoding
                    lastSequence.litlen = q_litlen;
                    lastSequence.mlen = maxML;
                    lastSequence.off = maxOffset;
                    cur = 0;
                    last_pos = lastSequence.litlen + lastSequence.mlen;
                    state = ST_CHUNK_END;
                    break;
                }
                pr_literalsPrice = (U32)opt[0].price + litLengthPrice(0, w, optLevel);
                for (U32 pos = 1; pos < minMatch; pos++)
                    opt[pos].price = MAX_PRICE;
                pr_pos = minMatch;
                pr_matchNb = 0;
                ZS_M_GET(0, pm_off, pm_len);
                state = ST_PRICE_FIRST;
                break;
            }
            if (!nbMatches) {
                state = ST_CUR_NEXT;
                break;
            }
            {
                const U32 maxML = last_m_len;
                if ((maxML > sufficient_len) || (cur + maxML >= OPT_NUM)) {
                    lastSequence.mlen = maxML;
                    lastSequence.off = last_m_off;
                    lastSequence.litlen = q_litlen;
                    cur -= cur_litlen_back; // last sequence is actually only literals (may underflow)
                    last_pos = cur + lastSequence.litlen + lastSequence.mlen;
                    if (cur > OPT_NUM)
                        cur = 0; // underflow => first match
                    state = ST_CHUNK_END;
                    break;
                }
            }
            pr_matchNb = 0;
            ZS_M_GET(0, pm_off, pm_len);
            pm_start = minMatch;
            pr_pos = pm_len; // mlen cursor of the downward scan
            state = ST_PRICE_CUR;
        } while (0);
        if (state == ST_PRICE_FIRST) do {
            // for (matchNb...) for ( ; pos <= end ; pos++ ): a few positions per micro-step
            U32 budget = SM_PRICE_STEPS;
            while (budget && pr_matchNb < nbMatches) {
                if (pr_pos <= pm_len) {
                    const U32 sequencePrice = pr_literalsPrice + getMatchPrice(pm_off, pr_pos, w, optLevel);
                    Optimal o; // (rep is set when the forward pass reaches the position)
                    o.price = (int)sequencePrice;
                    o.off = pm_off;
                    o.mlen = pr_pos;
                    o.litlen = q_litlen;
                    storeHead(opt[pr_pos], o);
                    pr_pos++;
                    budget--;
                } else {
                    pr_matchNb++;
                    if (pr_matchNb < nbMatches) {
                        ZS_M_GET(pr_matchNb, pm_off, pm_len);
                    }
                }
            }
            if (pr_matchNb >= nbMatches) {
                last_pos = pr_pos - 1;
                cur = 1;
                state = ST_CUR_BEGIN; // last_pos >= minMatch: the loop body runs
            }
        } while (0);
        if (state == ST_PRICE_CUR) do {
            // for (matchNb...) for (mlen = lastML; mlen >= startML; mlen--): a few lengths per micro-step.  The prices the lengths
            // of one batch compete with are read TOGETHER before the batch (one round trip instead of one per length): every
            // position is visited once per `cur`, the fills below only touch positions beyond last_pos, so a price read up front
            // is still the one in memory when its turn comes -- and a position beyond the last_pos of the batch's start holds
            // MAX_PRICE by then (or is still beyond last_pos, where the comparison is not made at all).
            U32 budget = SM_PRICE_STEPS;
            while (budget && pr_matchNb < nbMatches) {
                bool next = false;
                if (pr_pos >= pm_start) {
                    const U32 lp0 = last_pos;
                    const U32 avail = pr_pos - pm_start + 1;
                    const U32 nb = budget < avail ? budget : avail;
                    int pp[SM_PRICE_STEPS];
#pragma unroll
                    for (U32 t = 0; t < SM_PRICE_STEPS; ++t) {
                        const U32 pos = cur + pr_pos - t;
                        pp[t] = (t < nb && pos <= lp0) ? opt[pos].price : MAX_PRICE;
                    }
#pragma unroll
                    for (U32 t = 0; t < SM_PRICE_STEPS; ++t) {
                        if (t < nb && !next) {
                            const U32 mlen = pr_pos;
                            const U32 pos = cur + mlen;
                            const int price = (int)(basePrice + getMatchPrice(pm_off, mlen, w, optLevel));
                            if ((pos > last_pos) || (price < pp[t])) {
                                while (last_pos < pos) {
                                    opt[last_pos + 1].price = MAX_PRICE;
                                    last_pos++;
                                }
                                Optimal o;
                                o.price = price;
                                o.off = pm_off;
                                o.mlen = mlen;
                                o.litlen = q_litlen;
                                storeHead(opt[pos], o);
                            } else if (optLevel == 0)
                                next = true; // early update abort
                            pr_pos--;
                            budget--;
                        }
                    }
                    if (pr_pos < pm_start)
                        next = true;
                } else
                    next = true;
                if (next) {
                    pr_matchNb++;
                    if (pr_matchNb < nbMatches) {
                        pm_start = pm_len + 1; // matches[matchNb - 1].len + 1
                        ZS_M_GET(pr_matchNb, pm_off, pm_len);
                        pr_pos = pm_len;
                    }
                }
            }
            if (pr_matchNb >= nbMatches)
                state = ST_CUR_NEXT;
        } while (0);
        if (state == ST_CHUNK_END) do {
            if (lastSequence.mlen != 0) {
                U32 reps[3];
                updateRep(reps, opt[cur].rep, lastSequence.off, lastSequence.litlen == 0);
                rep[0] = reps[0];
                rep[1] = reps[1];
                rep[2] = reps[2];
            } else {
                rep[0] = opt[cur].rep[0];
                rep[1] = opt[cur].rep[1];
                rep[2] = opt[cur].rep[2];
            }
            storeEnd = cur + 1;
            U32 storeStart = storeEnd;
            U32 seqPos = cur;
            opt[storeEnd] = lastSequence;
            while (seqPos > 0) {
                const U32 backDist = opt[seqPos].litlen + opt[seqPos].mlen;
                storeStart--;
                opt[storeStart] = opt[seqPos];
                seqPos = (seqPos > backDist) ? seqPos - backDist : 0;
            }
            storePos = storeStart;
            state = ST_STORE;
        } while (0);
    }
    return srcSize - anchor;
#undef ZS_M_SET
#undef ZS_M_GET
}

} // namespace zs
