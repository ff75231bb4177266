// Derived from Zstandard 1.4.9 (Copyright (c) 2016-present, Facebook, Inc.; BSD license): see NOTICE in this directory.
// zs_opt.h -- binary-tree match finder + optimal parser of zstd 1.4.9 (lib/compress/zstd_opt.c), strategies btopt /
// btultra / btultra2, no dictionary, one block.  Follows the library decision by decision: same hash functions, same tree
// updates, same price model (fractional-bit weights, BITCOST_ACCURACY 8), same statistics updates, same tie-breaks.
//
// Positions are "indices" as the library keeps them: index = offset in the source + idx0, where idx0 = 1 for the first
// pass (ZSTD_window_init leaves index 0 invalid and ZSTD_window_update puts the first byte at dictLimit = 1) and
// idx0 = 1 + srcSize for the second pass of btultra2 (ZSTD_initStats_ultra moves window.base back by srcSize).
#pragma once
#include "zs_common.h"

namespace zs {

constexpr U32 BITCOST_ACCURACY = 8;
constexpr U32 BITCOST_MULTIPLIER = 1u << BITCOST_ACCURACY;
constexpr U32 LITFREQ_ADD = 2;
constexpr U32 FREQ_DIV = 4;
constexpr int MAX_PRICE = 1 << 30;
constexpr U32 PREDEF_THRESHOLD = 1024;

enum { zop_dynamic = 0, zop_predef = 1 };

// per-frame working memory (all pointers into the frame's workspace)
struct OptWs {
    // match state
    U32 *hashTable, *hashTable3, *chainTable;
    U32 hashLog3;
    U32 idx0;          // index of src[0]
    U32 dictLimit;     // == lowLimit (no dictionary)
    U32 nextToUpdate;
    CParams cp;
    // optimal parser
    Optimal *opt;      // OPT_NUM + 1 entries
    Match *matches;    // OPT_NUM + 1 entries
    U32 *bk;           // (32 bits: a run of literals can be as long as the block)
    U16 *fw;           // group parser (zs_opt_grp.h): per position of the chunk, the length of the sequence ending there / the position
                       // the next sequence of the chosen path ends at (OPT_NUM + 2 entries each)
    Match *fastMatches; // the first fastMatchCap matches of a request live here instead (fast memory: LDS on the device)
    U32 fastMatchCap;
    U32 *litFreq;      // 256
    U32 *litLengthFreq; // 36
    U32 *matchLengthFreq; // 53
    U32 *offCodeFreq;  // 32
    U32 litSum, litLengthSum, matchLengthSum, offCodeSum;
    U32 litSumBasePrice, litLengthSumBasePrice, matchLengthSumBasePrice, offCodeSumBasePrice;
    U32 priceType;
    // sequence store
    Seq *seqs;
    U32 nSeq;
    BYTE *lits;
    U32 nLits;
};

// ---- hashes (zstd_compress_internal.h) ----
ZFN U32 hash3(U32 u, U32 h) { return ((u << 8) * 506832829u) >> (32 - h); }
ZFN U32 hash4(U32 u, U32 h) { return (u * 2654435761u) >> (32 - h); }
ZFN U32 hash5(U64 u, U32 h) { return (U32)(((u << 24) * 889523592379ULL) >> (64 - h)); }
ZFN U32 hash6(U64 u, U32 h) { return (U32)(((u << 16) * 227718039650203ULL) >> (64 - h)); }
ZFN U32 hashPtr(const BYTE *p, U32 hBits, U32 mls)
{
    switch (mls) {
    default:
    case 4: return hash4(read32(p), hBits);
    case 5: return hash5(read64(p), hBits);
    case 6: return hash6(read64(p), hBits);
    }
}

// ---- price model ----
ZFN U32 bitWeight(U32 stat) { return highbit32(stat + 1) * BITCOST_MULTIPLIER; }
ZFN U32 fracWeight(U32 rawStat)
{
    const U32 stat = rawStat + 1;
    const U32 hb = highbit32(stat);
    const U32 BWeight = hb * BITCOST_MULTIPLIER;
    const U32 FWeight = (stat << BITCOST_ACCURACY) >> hb;
    return BWeight + FWeight;
}
ZFN U32 weight(U32 stat, int optLevel) { return optLevel ? fracWeight(stat) : bitWeight(stat); }

ZFN void setBasePrices(OptWs &w, int optLevel)
{
    w.litSumBasePrice = weight(w.litSum, optLevel);
    w.litLengthSumBasePrice = weight(w.litLengthSum, optLevel);
    w.matchLengthSumBasePrice = weight(w.matchLengthSum, optLevel);
    w.offCodeSumBasePrice = weight(w.offCodeSum, optLevel);
}

ZFN U32 downscaleStat(U32 *table, U32 lastEltIndex, int malus)
{
    U32 sum = 0;
    for (U32 s = 0; s < lastEltIndex + 1; s++) {
        table[s] = 1 + (table[s] >> (FREQ_DIV + malus));
        sum += table[s];
    }
    return sum;
}

// ZSTD_rescaleFreqs (no dictionary: the literal statistics of a first block come from the block itself)
ZFN void rescaleFreqs(OptWs &w, const BYTE *src, U32 srcSize, int optLevel)
{
    w.priceType = zop_dynamic;
    if (w.litLengthSum == 0) { // first block
        if (srcSize <= PREDEF_THRESHOLD)
            w.priceType = zop_predef;
        for (U32 s = 0; s <= MaxLit; ++s)
            w.litFreq[s] = 0;
        for (U32 i = 0; i < srcSize; ++i) // HIST_count_simple
            w.litFreq[src[i]]++;
        w.litSum = downscaleStat(w.litFreq, MaxLit, 1);
        for (U32 ll = 0; ll <= MaxLL; ll++)
            w.litLengthFreq[ll] = 1;
        w.litLengthSum = MaxLL + 1;
        for (U32 ml = 0; ml <= MaxML; ml++)
            w.matchLengthFreq[ml] = 1;
        w.matchLengthSum = MaxML + 1;
        for (U32 of = 0; of <= MaxOff; of++)
            w.offCodeFreq[of] = 1;
        w.offCodeSum = MaxOff + 1;
    } else { // new block: previous statistics, scaled down
        w.litSum = downscaleStat(w.litFreq, MaxLit, 1);
        w.litLengthSum = downscaleStat(w.litLengthFreq, MaxLL, 0);
        w.matchLengthSum = downscaleStat(w.matchLengthFreq, MaxML, 0);
        w.offCodeSum = downscaleStat(w.offCodeFreq, MaxOff, 0);
    }
    setBasePrices(w, optLevel);
}

ZFN U32 upscaleStat(U32 *table, U32 lastEltIndex, int bonus)
{
    U32 sum = 0;
    for (U32 s = 0; s < lastEltIndex + 1; s++) {
        table[s] <<= FREQ_DIV + bonus;
        table[s]--;
        sum += table[s];
    }
    return sum;
}

ZFN void upscaleStats(OptWs &w)
{
    w.litSum = upscaleStat(w.litFreq, MaxLit, 0);
    w.litLengthSum = upscaleStat(w.litLengthFreq, MaxLL, 0);
    w.matchLengthSum = upscaleStat(w.matchLengthFreq, MaxML, 0);
    w.offCodeSum = upscaleStat(w.offCodeFreq, MaxOff, 0);
}

ZFN U32 rawLiteralsCost(const BYTE *literals, U32 litLength, const OptWs &w, int optLevel)
{
    if (litLength == 0)
        return 0;
    if (w.priceType == zop_predef)
        return (litLength * 6) * BITCOST_MULTIPLIER;
    U32 price = litLength * w.litSumBasePrice;
    for (U32 u = 0; u < litLength; u++)
        price -= weight(w.litFreq[literals[u]], optLevel);
    return price;
}

ZFN U32 litLengthPrice(U32 litLength, const OptWs &w, int optLevel)
{
    if (w.priceType == zop_predef)
        return weight(litLength, optLevel);
    const U32 llCode = LLcode(litLength);
    return (LLbits(llCode) * BITCOST_MULTIPLIER) + w.litLengthSumBasePrice - weight(w.litLengthFreq[llCode], optLevel);
}

ZFN U32 getMatchPrice(U32 offset, U32 matchLength, const OptWs &w, int optLevel)
{
    const U32 offCode = highbit32(offset + 1);
    const U32 mlBase = matchLength - MINMATCH;
    if (w.priceType == zop_predef)
        return weight(mlBase, optLevel) + ((16 + offCode) * BITCOST_MULTIPLIER);
    U32 price = (offCode * BITCOST_MULTIPLIER) + (w.offCodeSumBasePrice - weight(w.offCodeFreq[offCode], optLevel));
    if ((optLevel < 2) && offCode >= 20)
        price += (offCode - 19) * 2 * BITCOST_MULTIPLIER;
    const U32 mlCode = MLcode(mlBase);
    price += (MLbits(mlCode) * BITCOST_MULTIPLIER) + (w.matchLengthSumBasePrice - weight(w.matchLengthFreq[mlCode], optLevel));
    price += BITCOST_MULTIPLIER / 5;
    return price;
}

ZFN void updateStats(OptWs &w, U32 litLength, const BYTE *literals, U32 offsetCode, U32 matchLength)
{
    for (U32 u = 0; u < litLength; u++)
        w.litFreq[literals[u]] += LITFREQ_ADD;
    w.litSum += litLength * LITFREQ_ADD;
    w.litLengthFreq[LLcode(litLength)]++;
    w.litLengthSum++;
    w.offCodeFreq[highbit32(offsetCode + 1)]++;
    w.offCodeSum++;
    w.matchLengthFreq[MLcode(matchLength - MINMATCH)]++;
    w.matchLengthSum++;
}

// ---- binary tree ----
// ZSTD_insertBt1: inserts the position `current` (index) into the tree, returns how many positions may be skipped
ZFN U32 insertBt1(OptWs &w, const BYTE *src, U32 current, const BYTE *iend, U32 mls)
{
    const CParams &cp = w.cp;
    const BYTE *const ip = src + (current - w.idx0);
    const U32 h = hashPtr(ip, cp.hashLog, mls);
    U32 *const bt = w.chainTable;
    const U32 btLog = cp.chainLog - 1;
    const U32 btMask = (1u << btLog) - 1;
    U32 matchIndex = w.hashTable[h];
    U32 commonLengthSmaller = 0, commonLengthLarger = 0;
    const U32 btLow = btMask >= current ? 0 : current - btMask;
    U32 *smallerPtr = bt + 2 * (current & btMask);
    U32 *largerPtr = smallerPtr + 1;
    U32 dummy32;
    const U32 windowLow = w.dictLimit;
    U32 matchEndIdx = current + 8 + 1;
    U32 bestLength = 8;
    U32 nbCompares = 1u << cp.searchLog;

    w.hashTable[h] = current;

    while (nbCompares-- && (matchIndex >= windowLow)) {
        U32 *const nextPtr = bt + 2 * (matchIndex & btMask);
        U32 matchLength = commonLengthSmaller < commonLengthLarger ? commonLengthSmaller : commonLengthLarger;
        const BYTE *const match = src + (matchIndex - w.idx0);
        matchLength += count(ip + matchLength, match + matchLength, iend);

        if (matchLength > bestLength) {
            bestLength = matchLength;
            if (matchLength > matchEndIdx - matchIndex)
                matchEndIdx = matchIndex + matchLength;
        }
        if (ip + matchLength == iend)
            break; // equal: no way to know if inf or sup

        if (match[matchLength] < ip[matchLength]) {
            *smallerPtr = matchIndex;
            commonLengthSmaller = matchLength;
            if (matchIndex <= btLow) {
                smallerPtr = &dummy32;
                break;
            }
            smallerPtr = nextPtr + 1;
            matchIndex = nextPtr[1];
        } else {
            *largerPtr = matchIndex;
            commonLengthLarger = matchLength;
            if (matchIndex <= btLow) {
                largerPtr = &dummy32;
                break;
            }
            largerPtr = nextPtr;
            matchIndex = nextPtr[0];
        }
    }
    *smallerPtr = *largerPtr = 0;
    {
        U32 positions = 0;
        if (bestLength > 384)
            positions = bestLength - 384 < 192 ? bestLength - 384 : 192;
        const U32 adv = matchEndIdx - (current + 8);
        return positions > adv ? positions : adv;
    }
}

ZFN void updateTree(OptWs &w, const BYTE *src, U32 target, const BYTE *iend, U32 mls)
{
    U32 idx = w.nextToUpdate;
    while (idx < target)
        idx += insertBt1(w, src, idx, iend, mls);
    w.nextToUpdate = target;
}

ZFN U32 insertAndFindFirstIndexHash3(OptWs &w, const BYTE *src, U32 *nextToUpdate3, U32 target)
{
    U32 *const hashTable3 = w.hashTable3;
    const U32 hashLog3 = w.hashLog3;
    U32 idx = *nextToUpdate3;
    const U32 h3 = hash3(read32(src + (target - w.idx0)), hashLog3);
    while (idx < target) {
        hashTable3[hash3(read32(src + (idx - w.idx0)), hashLog3)] = idx;
        idx++;
    }
    *nextToUpdate3 = target;
    return hashTable3[h3];
}

ZFN U32 readMINMATCH(const BYTE *p, U32 length)
{
    return length == 3 ? (read32(p) << 8) : read32(p);
}

// ZSTD_insertBtAndGetAllMatches (noDict)
ZFN U32 insertBtAndGetAllMatches(Match *matches, OptWs &w, const BYTE *src, U32 *nextToUpdate3, U32 current, const BYTE *iLimit,
                                 const U32 rep[3], U32 ll0, U32 lengthToBeat, U32 mls)
{
    const CParams &cp = w.cp;
    const U32 sufficient_len = cp.targetLength < OPT_NUM - 1 ? cp.targetLength : OPT_NUM - 1;
    const BYTE *const ip = src + (current - w.idx0);
    const U32 minMatch = (mls == 3) ? 3 : 4;
    const U32 h = hashPtr(ip, cp.hashLog, mls);
    U32 matchIndex = w.hashTable[h];
    U32 *const bt = w.chainTable;
    const U32 btLog = cp.chainLog - 1;
    const U32 btMask = (1u << btLog) - 1;
    U32 commonLengthSmaller = 0, commonLengthLarger = 0;
    const U32 dictLimit = w.dictLimit;
    const U32 btLow = (btMask >= current) ? 0 : current - btMask;
    // ZSTD_getLowestMatchIndex (no dictionary)
    const U32 maxDistance = 1u << cp.windowLog;
    const U32 lowestValid = w.dictLimit;
    const U32 windowLow = (current - lowestValid > maxDistance) ? current - maxDistance : lowestValid;
    const U32 matchLow = windowLow ? windowLow : 1;
    U32 *smallerPtr = bt + 2 * (current & btMask);
    U32 *largerPtr = bt + 2 * (current & btMask) + 1;
    U32 matchEndIdx = current + 8 + 1;
    U32 dummy32;
    U32 mnum = 0;
    U32 nbCompares = 1u << cp.searchLog;
    U32 bestLength = lengthToBeat - 1;

    // repcodes
    {
        const U32 lastR = REP_NUM + ll0;
        for (U32 repCode = ll0; repCode < lastR; repCode++) {
            const U32 repOffset = (repCode == REP_NUM) ? (rep[0] - 1) : rep[repCode];
            const U32 repIndex = current - repOffset;
            U32 repLen = 0;
            if (repOffset - 1 /* intentional overflow, discards 0 and -1 */ < current - dictLimit) {
                if ((repIndex >= windowLow) & (readMINMATCH(ip, minMatch) == readMINMATCH(ip - repOffset, minMatch)))
                    repLen = count(ip + minMatch, ip + minMatch - repOffset, iLimit) + minMatch;
            }
            if (repLen > bestLength) {
                bestLength = repLen;
                matches[mnum].off = repCode - ll0;
                matches[mnum].len = repLen;
                mnum++;
                if ((repLen > sufficient_len) | (ip + repLen == iLimit))
                    return mnum;
            }
        }
    }

    // HC3 match finder
    if ((mls == 3) && (bestLength < mls)) {
        const U32 matchIndex3 = insertAndFindFirstIndexHash3(w, src, nextToUpdate3, current);
        if ((matchIndex3 >= matchLow) & (current - matchIndex3 < (1u << 18))) {
            const BYTE *const match = src + (matchIndex3 - w.idx0);
            const U32 mlen = count(ip, match, iLimit);
            if (mlen >= mls) {
                bestLength = mlen;
                matches[0].off = (current - matchIndex3) + REP_MOVE;
                matches[0].len = mlen;
                mnum = 1;
                if ((mlen > sufficient_len) | (ip + mlen == iLimit)) {
                    w.nextToUpdate = current + 1; // skip insertion
                    return 1;
                }
            }
        }
    }

    w.hashTable[h] = current;

    while (nbCompares-- && (matchIndex >= matchLow)) {
        U32 *const nextPtr = bt + 2 * (matchIndex & btMask);
        U32 matchLength = commonLengthSmaller < commonLengthLarger ? commonLengthSmaller : commonLengthLarger;
        const BYTE *const match = src + (matchIndex - w.idx0);
        matchLength += count(ip + matchLength, match + matchLength, iLimit);

        if (matchLength > bestLength) {
            if (matchLength > matchEndIdx - matchIndex)
                matchEndIdx = matchIndex + matchLength;
            bestLength = matchLength;
            matches[mnum].off = (current - matchIndex) + REP_MOVE;
            matches[mnum].len = matchLength;
            mnum++;
            if ((matchLength > OPT_NUM) | (ip + matchLength == iLimit))
                break; // drop, to preserve bt consistency
        }

        if (match[matchLength] < ip[matchLength]) {
            *smallerPtr = matchIndex;
            commonLengthSmaller = matchLength;
            if (matchIndex <= btLow) {
                smallerPtr = &dummy32;
                break;
            }
            smallerPtr = nextPtr + 1;
            matchIndex = nextPtr[1];
        } else {
            *largerPtr = matchIndex;
            commonLengthLarger = matchLength;
            if (matchIndex <= btLow) {
                largerPtr = &dummy32;
                break;
            }
            largerPtr = nextPtr;
            matchIndex = nextPtr[0];
        }
    }
    *smallerPtr = *largerPtr = 0;
    w.nextToUpdate = matchEndIdx - 8; // skip repetitive patterns
    return mnum;
}

ZFN U32 btGetAllMatches(Match *matches, OptWs &w, const BYTE *src, U32 *nextToUpdate3, U32 current, const BYTE *iHighLimit,
                        const U32 rep[3], U32 ll0, U32 lengthToBeat)
{
    const U32 mls = w.cp.minMatch;
    if (current < w.nextToUpdate)
        return 0; // skipped area
    const U32 m = mls <= 3 ? 3 : (mls == 4 ? 4 : (mls == 5 ? 5 : 6));
    updateTree(w, src, current, iHighLimit, m);
    return insertBtAndGetAllMatches(matches, w, src, nextToUpdate3, current, iHighLimit, rep, ll0, lengthToBeat, m);
}

// ZSTD_updateRep
ZFN void updateRep(U32 out[3], const U32 rep[3], U32 offset, U32 ll0)
{
    if (offset >= REP_NUM) {
        const U32 r0 = rep[0], r1 = rep[1];
        out[2] = r1;
        out[1] = r0;
        out[0] = offset - REP_MOVE;
    } else {
        const U32 repCode = offset + ll0;
        if (repCode > 0) {
            const U32 currentOffset = (repCode == REP_NUM) ? (rep[0] - 1) : rep[repCode];
            const U32 r0 = rep[0], r1 = rep[1], r2 = rep[2];
            out[2] = (repCode >= 2) ? r1 : r2;
            out[1] = r0;
            out[0] = currentOffset;
        } else {
            const U32 r0 = rep[0], r1 = rep[1], r2 = rep[2];
            out[0] = r0;
            out[1] = r1;
            out[2] = r2;
        }
    }
}

ZFN void storeSeq(OptWs &w, U32 litLength, const BYTE *literals, U32 offCode, U32 matchLength)
{
    for (U32 i = 0; i < litLength; ++i)
        w.lits[w.nLits + i] = literals[i];
    w.nLits += litLength;
    Seq &s = w.seqs[w.nSeq++];
    s.offCode = offCode;
    s.litLength = litLength;
    s.matchLength = matchLength;
}

// ZSTD_compressBlock_opt_generic (noDict); returns the size of the last literals run
ZFN U32 compressBlockOpt(OptWs &w, U32 rep[3], const BYTE *src, U32 srcSize, int optLevel)
{
    const BYTE *const istart = src;
    const BYTE *ip = istart;
    const BYTE *anchor = istart;
    const BYTE *const iend = istart + srcSize;
    const BYTE *const ilimit = iend - 8;
    const CParams &cp = w.cp;
    const U32 sufficient_len = cp.targetLength < OPT_NUM - 1 ? cp.targetLength : OPT_NUM - 1;
    const U32 minMatch = (cp.minMatch == 3) ? 3 : 4;
    U32 nextToUpdate3 = w.nextToUpdate;
    Optimal *const opt = w.opt;
    Match *const matches = w.matches;
    Optimal lastSequence;
    lastSequence.price = 0;
    lastSequence.off = lastSequence.mlen = lastSequence.litlen = 0;
    lastSequence.rep[0] = lastSequence.rep[1] = lastSequence.rep[2] = 0;

    rescaleFreqs(w, src, srcSize, optLevel);
    ip += (w.idx0 == w.dictLimit); // ip == prefixStart (always the case here: the block starts the window)

    while (ip < ilimit) {
        U32 cur, last_pos = 0;
        bool shortest = false;

        // find first match
        {
            const U32 litlen = (U32)(ip - anchor);
            const U32 ll0 = !litlen;
            const U32 nbMatches = btGetAllMatches(matches, w, src, &nextToUpdate3, (U32)(ip - src) + w.idx0, iend, rep, ll0, minMatch);
            if (!nbMatches) {
                ip++;
                continue;
            }
            for (U32 i = 0; i < REP_NUM; i++)
                opt[0].rep[i] = rep[i];
            opt[0].mlen = 0;
            opt[0].litlen = litlen;
            opt[0].price = (int)litLengthPrice(litlen, w, optLevel);

            {
                const U32 maxML = matches[nbMatches - 1].len;
                const U32 maxOffset = matches[nbMatches - 1].off;
                if (maxML > sufficient_len) {
                    lastSequence.litlen = litlen;
                    lastSequence.mlen = maxML;
                    lastSequence.off = maxOffset;
                    cur = 0;
                    last_pos = lastSequence.litlen + lastSequence.mlen;
                    shortest = true;
                }
            }
            if (!shortest) {
                const U32 literalsPrice = (U32)opt[0].price + litLengthPrice(0, w, optLevel);
                U32 pos;
                for (pos = 1; pos < minMatch; pos++)
                    opt[pos].price = MAX_PRICE;
                for (U32 matchNb = 0; matchNb < nbMatches; matchNb++) {
                    const U32 offset = matches[matchNb].off;
                    const U32 end = matches[matchNb].len;
                    for (; pos <= end; pos++) {
                        const U32 matchPrice = getMatchPrice(offset, pos, w, optLevel);
                        const U32 sequencePrice = literalsPrice + matchPrice;
                        opt[pos].mlen = pos;
                        opt[pos].off = offset;
                        opt[pos].litlen = litlen;
                        opt[pos].price = (int)sequencePrice;
                    }
                }
                last_pos = pos - 1;
            }
        }

        // check further positions
        if (!shortest) {
            for (cur = 1; cur <= last_pos; cur++) {
                const BYTE *const inr = ip + cur;
                {
                    const U32 litlen = (opt[cur - 1].mlen == 0) ? opt[cur - 1].litlen + 1 : 1;
                    const int price = opt[cur - 1].price + (int)rawLiteralsCost(ip + cur - 1, 1, w, optLevel) +
                                      (int)litLengthPrice(litlen, w, optLevel) - (int)litLengthPrice(litlen - 1, w, optLevel);
                    if (price <= opt[cur].price) {
                        opt[cur].mlen = 0;
                        opt[cur].off = 0;
                        opt[cur].litlen = litlen;
                        opt[cur].price = price;
                    }
                }
                // repcodes of the current position
                if (opt[cur].mlen != 0) {
                    const U32 prev = cur - opt[cur].mlen;
                    updateRep(opt[cur].rep, opt[prev].rep, opt[cur].off, opt[cur].litlen == 0);
                } else {
                    for (U32 i = 0; i < REP_NUM; ++i)
                        opt[cur].rep[i] = opt[cur - 1].rep[i];
                }

                if (inr > ilimit)
                    continue; // last match must start at a minimum distance of 8 from oend
                if (cur == last_pos)
                    break;
                if ((optLevel == 0) && (opt[cur + 1].price <= opt[cur].price + (int)(BITCOST_MULTIPLIER / 2)))
                    continue; // skip unpromising positions

                {
                    const U32 ll0 = (opt[cur].mlen != 0);
                    const U32 litlen = (opt[cur].mlen == 0) ? opt[cur].litlen : 0;
                    const U32 previousPrice = (U32)opt[cur].price;
                    const U32 basePrice = previousPrice + litLengthPrice(0, w, optLevel);
                    const U32 nbMatches = btGetAllMatches(matches, w, src, &nextToUpdate3, (U32)(inr - src) + w.idx0, iend, opt[cur].rep, ll0, minMatch);
                    if (!nbMatches)
                        continue;
                    {
                        const U32 maxML = matches[nbMatches - 1].len;
                        if ((maxML > sufficient_len) || (cur + maxML >= OPT_NUM)) {
                            lastSequence.mlen = maxML;
                            lastSequence.off = matches[nbMatches - 1].off;
                            lastSequence.litlen = litlen;
                            cur -= (opt[cur].mlen == 0) ? opt[cur].litlen : 0; // last sequence is actually only literals (may underflow)
                            last_pos = cur + lastSequence.litlen + lastSequence.mlen;
                            if (cur > OPT_NUM)
                                cur = 0; // underflow => first match
                            shortest = true;
                            break;
                        }
                    }
                    for (U32 matchNb = 0; matchNb < nbMatches; matchNb++) {
                        const U32 offset = matches[matchNb].off;
                        const U32 lastML = matches[matchNb].len;
                        const U32 startML = (matchNb > 0) ? matches[matchNb - 1].len + 1 : minMatch;
                        for (U32 mlen = lastML; mlen >= startML; mlen--) { // scan downward
                            const U32 pos = cur + mlen;
                            const int price = (int)(basePrice + getMatchPrice(offset, mlen, w, optLevel));
                            if ((pos > last_pos) || (price < opt[pos].price)) {
                                while (last_pos < pos) {
                                    opt[last_pos + 1].price = MAX_PRICE;
                                    last_pos++;
                                }
                                opt[pos].mlen = mlen;
                                opt[pos].off = offset;
                                opt[pos].litlen = litlen;
                                opt[pos].price = price;
                            } else {
                                if (optLevel == 0)
                                    break; // early update abort
                            }
                        }
                    }
                }
            }
            if (!shortest) {
                lastSequence = opt[last_pos];
                const U32 tl = lastSequence.litlen + lastSequence.mlen;
                cur = last_pos > tl ? last_pos - tl : 0; // single sequence, and it starts before `ip`
            }
        }

        // _shortestPath: cur, last_pos, lastSequence are set
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
        {
            const U32 storeEnd = cur + 1;
            U32 storeStart = storeEnd;
            U32 seqPos = cur;
            opt[storeEnd] = lastSequence;
            while (seqPos > 0) {
                const U32 backDist = opt[seqPos].litlen + opt[seqPos].mlen;
                storeStart--;
                opt[storeStart] = opt[seqPos];
                seqPos = (seqPos > backDist) ? seqPos - backDist : 0;
            }
            for (U32 storePos = storeStart; storePos <= storeEnd; storePos++) {
                const U32 llen = opt[storePos].litlen;
                const U32 mlen = opt[storePos].mlen;
                const U32 offCode = opt[storePos].off;
                const U32 advance = llen + mlen;
                if (mlen == 0) { // only literals => must be last "sequence", actually starting a new stream of sequences
                    ip = anchor + llen;
                    continue;
                }
                updateStats(w, llen, anchor, offCode, mlen);
                storeSeq(w, llen, anchor, offCode, mlen);
                anchor += advance;
                ip = anchor;
            }
            setBasePrices(w, optLevel);
        }
    }
    return (U32)(iend - anchor);
}

// the block compressor ZSTD_selectBlockCompressor picks for btopt / btultra / btultra2 at the start of a frame;
// fills w.seqs / w.lits (without the last literals) and returns their count in *lastLits
// (PARSE = compressBlockOpt, the loop nest, or compressBlockOptSM of zs_opt_sm.h, the same parse as one loop of micro-steps)
template <typename PARSE> ZFN void compressBlockBt(OptWs &w, U32 rep[3], const BYTE *src, U32 srcSize, U32 *lastLits, PARSE parse)
{
    w.litSum = w.litLengthSum = w.matchLengthSum = w.offCodeSum = 0;
    w.nSeq = 0;
    w.nLits = 0;
    w.idx0 = 1;
    w.dictLimit = 1;
    w.nextToUpdate = 1;
    w.hashLog3 = w.cp.minMatch == 3 ? (HASHLOG3_MAX < w.cp.windowLog ? HASHLOG3_MAX : w.cp.windowLog) : 0;
    if (w.cp.strategy == STRAT_BTULTRA2 && srcSize > PREDEF_THRESHOLD) {
        // ZSTD_initStats_ultra: a first pass collects statistics, its sequences are dropped
        U32 tmpRep[3] = {rep[0], rep[1], rep[2]};
        parse(w, tmpRep, src, srcSize, 2);
        w.nSeq = 0;
        w.nLits = 0;
        w.idx0 += srcSize;   // window.base -= srcSize
        w.dictLimit += srcSize;
        w.nextToUpdate = w.dictLimit;
        upscaleStats(w);
    }
    *lastLits = parse(w, rep, src, srcSize, w.cp.strategy == STRAT_BTOPT ? 0 : 2);
}

} // namespace zs
