// Derived from Zstandard 1.4.9 (Copyright (c) 2016-present, Facebook, Inc.; BSD license): see NOTICE in this directory.
// zs_entropy.h -- the entropy stage of a zstd block as libzstd 1.4.9 runs it for the first block of a frame
// (zstd_compress.c: ZSTD_entropyCompressSequences_internal; zstd_compress_literals.c; zstd_compress_sequences.c;
// huf_compress.c; fse_compress.c; hist.c): Huffman-coded literals (1 or 4 streams, tree description FSE-compressed or
// raw nibbles), three FSE-coded symbol streams with the library's choice between predefined / RLE / described tables,
// and every "not worth it" fallback (raw literals, RLE literals, raw block).
#pragma once
#include "zs_common.h"

namespace zs {

constexpr U32 LLFSELog = 9, MLFSELog = 9, OffFSELog = 8;
constexpr U32 FSE_MIN_TABLELOG = 5, FSE_MAX_TABLELOG = 12, FSE_DEFAULT_TABLELOG = 11;
constexpr U32 HUF_TABLELOG_MAX = 12, HUF_TABLELOG_DEFAULT = 11, HUF_SYMBOLVALUE_MAX = 255;
constexpr U32 DefaultMaxOff = 28;
constexpr U32 LONGNBSEQ = 0x7F00;

enum { set_basic = 0, set_rle = 1, set_compressed = 2, set_repeat = 3 };

ZCONST int16_t LL_defaultNorm[MaxLL + 1] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
ZCONST int16_t ML_defaultNorm[MaxML + 1] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                            1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
ZCONST int16_t OF_defaultNorm[DefaultMaxOff + 1] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};
constexpr U32 LL_defaultNormLog = 6, ML_defaultNormLog = 6, OF_defaultNormLog = 5;

// floor(-log2(x / 256) * 256), x in [1, 256); 0 for x = 0 (zstd_compress_sequences.c: kInverseProbabilityLog256)
ZCONST U16 kInverseProbabilityLog256[256] = {
    0, 2048, 1792, 1642, 1536, 1453, 1386, 1329, 1280, 1236, 1197, 1162, 1130, 1100, 1073, 1047, 1024, 1001, 980, 960, 941, 923,
    906, 889, 874, 859, 844, 830, 817, 804, 791, 779, 768, 756, 745, 734, 724, 714, 704, 694, 685, 676, 667, 658,
    650, 642, 633, 626, 618, 610, 603, 595, 588, 581, 574, 567, 561, 554, 548, 542, 535, 529, 523, 517, 512, 506,
    500, 495, 489, 484, 478, 473, 468, 463, 458, 453, 448, 443, 438, 434, 429, 424, 420, 415, 411, 407, 402, 398,
    394, 390, 386, 382, 377, 373, 370, 366, 362, 358, 354, 350, 347, 343, 339, 336, 332, 329, 325, 322, 318, 315,
    311, 308, 305, 302, 298, 295, 292, 289, 286, 282, 279, 276, 273, 270, 267, 264, 261, 258, 256, 253, 250, 247,
    244, 241, 239, 236, 233, 230, 228, 225, 222, 220, 217, 215, 212, 209, 207, 204, 202, 199, 197, 194, 192, 190,
    187, 185, 182, 180, 178, 175, 173, 171, 168, 166, 164, 162, 159, 157, 155, 153, 151, 149, 146, 144, 142, 140,
    138, 136, 134, 132, 130, 128, 126, 123, 121, 119, 117, 115, 114, 112, 110, 108, 106, 104, 102, 100, 98, 96,
    94, 93, 91, 89, 87, 85, 83, 82, 80, 78, 76, 74, 73, 71, 69, 67, 66, 64, 62, 61, 59, 57,
    55, 54, 52, 50, 49, 47, 46, 44, 42, 41, 39, 37, 36, 34, 33, 31, 30, 28, 26, 25, 23, 22,
    20, 19, 17, 16, 14, 13, 11, 10, 8, 7, 5, 4, 2, 1};

// ---- bit stream (bitstream.h: BIT_CStream_t): bits are appended LSB first, closed with a 1 bit ----
struct BitW {
    BYTE *start, *ptr;
    U64 acc;
    U32 nb;
};
ZFN void bitInit(BitW &b, BYTE *dst)
{
    b.start = b.ptr = dst;
    b.acc = 0;
    b.nb = 0;
}
ZFN void bitFlush(BitW &b)
{
    while (b.nb >= 8) {
        *b.ptr++ = (BYTE)b.acc;
        b.acc >>= 8;
        b.nb -= 8;
    }
}
ZFN void bitAdd(BitW &b, U32 value, U32 nbBits) // nbBits <= 31; at most 56 bits pending after a flush
{
    if (b.nb + nbBits > 64)
        bitFlush(b);
    b.acc |= (U64)(value & ((nbBits >= 32) ? 0xFFFFFFFFu : ((1u << nbBits) - 1u))) << b.nb;
    b.nb += nbBits;
}
ZFN U32 bitClose(BitW &b)
{
    bitAdd(b, 1, 1);
    bitFlush(b);
    if (b.nb) {
        *b.ptr++ = (BYTE)b.acc;
        b.nb = 0;
    }
    return (U32)(b.ptr - b.start);
}

// ---- FSE ----
struct FseSym {
    int deltaFindState;
    U32 deltaNbBits;
};
struct FseCTable { // sized for tableLog <= 9 and 53 symbols (sequence tables); the Huffman-weights table uses tableLog <= 6
    U16 tableLog, maxSymbolValue;
    U16 stateTable[512];
    FseSym symbolTT[MaxML + 1];
};

ZFN U32 fseMinTableLog(U32 srcSize, U32 maxSymbolValue)
{
    const U32 minBitsSrc = highbit32(srcSize) + 1;
    const U32 minBitsSymbols = highbit32(maxSymbolValue) + 2;
    return minBitsSrc < minBitsSymbols ? minBitsSrc : minBitsSymbols;
}

ZFN U32 fseOptimalTableLog(U32 maxTableLog, U32 srcSize, U32 maxSymbolValue, U32 minus)
{
    const U32 maxBitsSrc = highbit32(srcSize - 1) - minus;
    U32 tableLog = maxTableLog;
    const U32 minBits = fseMinTableLog(srcSize, maxSymbolValue);
    if (tableLog == 0)
        tableLog = FSE_DEFAULT_TABLELOG;
    if (maxBitsSrc < tableLog)
        tableLog = maxBitsSrc;
    if (minBits > tableLog)
        tableLog = minBits;
    if (tableLog < FSE_MIN_TABLELOG)
        tableLog = FSE_MIN_TABLELOG;
    if (tableLog > FSE_MAX_TABLELOG)
        tableLog = FSE_MAX_TABLELOG;
    return tableLog;
}

// FSE_normalizeM2; returns false on error
ZFN bool fseNormalizeM2(int16_t *norm, U32 tableLog, const U32 *count, U32 total, U32 maxSymbolValue, int16_t lowProbCount)
{
    const int16_t NOT_YET_ASSIGNED = -2;
    U32 s;
    U32 distributed = 0;
    U32 ToDistribute;
    const U32 lowThreshold = total >> tableLog;
    U32 lowOne = (U32)(((U64)total * 3) >> (tableLog + 1));

    for (s = 0; s <= maxSymbolValue; s++) {
        if (count[s] == 0) {
            norm[s] = 0;
            continue;
        }
        if (count[s] <= lowThreshold) {
            norm[s] = lowProbCount;
            distributed++;
            total -= count[s];
            continue;
        }
        if (count[s] <= lowOne) {
            norm[s] = 1;
            distributed++;
            total -= count[s];
            continue;
        }
        norm[s] = NOT_YET_ASSIGNED;
    }
    ToDistribute = (1u << tableLog) - distributed;
    if (ToDistribute == 0)
        return true;
    if ((total / ToDistribute) > lowOne) {
        lowOne = (U32)(((U64)total * 3) / (ToDistribute * 2));
        for (s = 0; s <= maxSymbolValue; s++) {
            if ((norm[s] == NOT_YET_ASSIGNED) && (count[s] <= lowOne)) {
                norm[s] = 1;
                distributed++;
                total -= count[s];
                continue;
            }
        }
        ToDistribute = (1u << tableLog) - distributed;
    }
    if (distributed == maxSymbolValue + 1) {
        U32 maxV = 0, maxC = 0;
        for (s = 0; s <= maxSymbolValue; s++)
            if (count[s] > maxC) {
                maxV = s;
                maxC = count[s];
            }
        norm[maxV] += (int16_t)ToDistribute;
        return true;
    }
    if (total == 0) {
        for (s = 0; ToDistribute > 0; s = (s + 1) % (maxSymbolValue + 1))
            if (norm[s] > 0) {
                ToDistribute--;
                norm[s]++;
            }
        return true;
    }
    {
        const U64 vStepLog = 62 - tableLog;
        const U64 mid = (1ULL << (vStepLog - 1)) - 1;
        const U64 rStep = ((((U64)1 << vStepLog) * ToDistribute) + mid) / (U32)total;
        U64 tmpTotal = mid;
        for (s = 0; s <= maxSymbolValue; s++) {
            if (norm[s] == NOT_YET_ASSIGNED) {
                const U64 end = tmpTotal + (count[s] * rStep);
                const U32 sStart = (U32)(tmpTotal >> vStepLog);
                const U32 sEnd = (U32)(end >> vStepLog);
                const U32 weight = sEnd - sStart;
                if (weight < 1)
                    return false;
                norm[s] = (int16_t)weight;
                tmpTotal = end;
            }
        }
    }
    return true;
}

// FSE_normalizeCount; returns tableLog, 0 for the rle special case, ~0u on error
ZFN U32 fseNormalizeCount(int16_t *normalizedCounter, U32 tableLog, const U32 *count, U32 total, U32 maxSymbolValue, bool useLowProbCount)
{
    if (tableLog == 0)
        tableLog = FSE_DEFAULT_TABLELOG;
    if (tableLog < FSE_MIN_TABLELOG || tableLog > FSE_MAX_TABLELOG)
        return ~0u;
    if (tableLog < fseMinTableLog(total, maxSymbolValue))
        return ~0u;
    {
        const U32 rtbTable[] = {0, 473195, 504333, 520860, 550000, 700000, 750000, 830000};
        const int16_t lowProbCount = useLowProbCount ? -1 : 1;
        const U64 scale = 62 - tableLog;
        const U64 step = ((U64)1 << 62) / (U32)total;
        const U64 vStep = 1ULL << (scale - 20);
        int stillToDistribute = 1 << tableLog;
        U32 largest = 0;
        int16_t largestP = 0;
        const U32 lowThreshold = total >> tableLog;

        for (U32 s = 0; s <= maxSymbolValue; s++) {
            if (count[s] == total)
                return 0; // rle special case
            if (count[s] == 0) {
                normalizedCounter[s] = 0;
                continue;
            }
            if (count[s] <= lowThreshold) {
                normalizedCounter[s] = lowProbCount;
                stillToDistribute--;
            } else {
                int16_t proba = (int16_t)((count[s] * step) >> scale);
                if (proba < 8) {
                    const U64 restToBeat = vStep * rtbTable[proba];
                    proba += (count[s] * step) - ((U64)proba << scale) > restToBeat;
                }
                if (proba > largestP) {
                    largestP = proba;
                    largest = s;
                }
                normalizedCounter[s] = proba;
                stillToDistribute -= proba;
            }
        }
        if (-stillToDistribute >= (normalizedCounter[largest] >> 1)) {
            if (!fseNormalizeM2(normalizedCounter, tableLog, count, total, maxSymbolValue, lowProbCount))
                return ~0u;
        } else
            normalizedCounter[largest] += (int16_t)stillToDistribute;
    }
    return tableLog;
}

// FSE_writeNCount (the buffer is always large enough here); returns the header size, 0 on error
ZFN U32 fseWriteNCount(BYTE *header, const int16_t *normalizedCounter, U32 maxSymbolValue, U32 tableLog)
{
    BYTE *out = header;
    int nbBits;
    const int tableSize = 1 << tableLog;
    int remaining;
    int threshold;
    U32 bitStream = 0;
    int bitCount = 0;
    U32 symbol = 0;
    const U32 alphabetSize = maxSymbolValue + 1;
    int previousIs0 = 0;

    bitStream += (tableLog - FSE_MIN_TABLELOG) << bitCount;
    bitCount += 4;
    remaining = tableSize + 1;
    threshold = tableSize;
    nbBits = (int)tableLog + 1;

    while ((symbol < alphabetSize) && (remaining > 1)) {
        if (previousIs0) {
            U32 start = symbol;
            while ((symbol < alphabetSize) && !normalizedCounter[symbol])
                symbol++;
            if (symbol == alphabetSize)
                break;
            while (symbol >= start + 24) {
                start += 24;
                bitStream += 0xFFFFU << bitCount;
                out[0] = (BYTE)bitStream;
                out[1] = (BYTE)(bitStream >> 8);
                out += 2;
                bitStream >>= 16;
            }
            while (symbol >= start + 3) {
                start += 3;
                bitStream += 3u << bitCount;
                bitCount += 2;
            }
            bitStream += (symbol - start) << bitCount;
            bitCount += 2;
            if (bitCount > 16) {
                out[0] = (BYTE)bitStream;
                out[1] = (BYTE)(bitStream >> 8);
                out += 2;
                bitStream >>= 16;
                bitCount -= 16;
            }
        }
        {
            int count = normalizedCounter[symbol++];
            const int max = (2 * threshold - 1) - remaining;
            remaining -= count < 0 ? -count : count;
            count++;
            if (count >= threshold)
                count += max;
            bitStream += (U32)count << bitCount;
            bitCount += nbBits;
            bitCount -= (count < max);
            previousIs0 = (count == 1);
            if (remaining < 1)
                return 0;
            while (remaining < threshold) {
                nbBits--;
                threshold >>= 1;
            }
        }
        if (bitCount > 16) {
            out[0] = (BYTE)bitStream;
            out[1] = (BYTE)(bitStream >> 8);
            out += 2;
            bitStream >>= 16;
            bitCount -= 16;
        }
    }
    if (remaining != 1)
        return 0;
    out[0] = (BYTE)bitStream;
    out[1] = (BYTE)(bitStream >> 8);
    out += (bitCount + 7) / 8;
    return (U32)(out - header);
}

// FSE_buildCTable_wksp; tableSymbol: scratch of 1 << tableLog bytes, cumul: maxSymbolValue + 2 words
ZFN void fseBuildCTable(FseCTable &ct, const int16_t *normalizedCounter, U32 maxSymbolValue, U32 tableLog, BYTE *tableSymbol, U32 *cumul)
{
    const U32 tableSize = 1u << tableLog;
    const U32 tableMask = tableSize - 1;
    const U32 step = (tableSize >> 1) + (tableSize >> 3) + 3;
    U32 highThreshold = tableSize - 1;
    ct.tableLog = (U16)tableLog;
    ct.maxSymbolValue = (U16)maxSymbolValue;
    cumul[0] = 0;
    for (U32 u = 1; u <= maxSymbolValue + 1; u++) {
        if (normalizedCounter[u - 1] == -1) {
            cumul[u] = cumul[u - 1] + 1;
            tableSymbol[highThreshold--] = (BYTE)(u - 1);
        } else
            cumul[u] = cumul[u - 1] + (U32)normalizedCounter[u - 1];
    }
    cumul[maxSymbolValue + 1] = tableSize + 1;
    {
        U32 position = 0;
        for (U32 symbol = 0; symbol <= maxSymbolValue; symbol++) {
            const int freq = normalizedCounter[symbol];
            for (int nbOccurrences = 0; nbOccurrences < freq; nbOccurrences++) {
                tableSymbol[position] = (BYTE)symbol;
                position = (position + step) & tableMask;
                while (position > highThreshold)
                    position = (position + step) & tableMask;
            }
        }
    }
    for (U32 u = 0; u < tableSize; u++) {
        const BYTE s = tableSymbol[u];
        ct.stateTable[cumul[s]++] = (U16)(tableSize + u);
    }
    {
        U32 total = 0;
        for (U32 s = 0; s <= maxSymbolValue; s++) {
            switch (normalizedCounter[s]) {
            case 0:
                ct.symbolTT[s].deltaNbBits = ((tableLog + 1) << 16) - (1u << tableLog);
                ct.symbolTT[s].deltaFindState = 0;
                break;
            case -1:
            case 1:
                ct.symbolTT[s].deltaNbBits = (tableLog << 16) - (1u << tableLog);
                ct.symbolTT[s].deltaFindState = (int)total - 1;
                total++;
                break;
            default: {
                const U32 maxBitsOut = tableLog - highbit32((U32)normalizedCounter[s] - 1);
                const U32 minStatePlus = (U32)normalizedCounter[s] << maxBitsOut;
                ct.symbolTT[s].deltaNbBits = (maxBitsOut << 16) - minStatePlus;
                ct.symbolTT[s].deltaFindState = (int)total - normalizedCounter[s];
                total += (U32)normalizedCounter[s];
            }
            }
        }
    }
}

ZFN void fseBuildCTableRle(FseCTable &ct, U32 symbolValue)
{
    ct.tableLog = 0;
    ct.maxSymbolValue = (U16)symbolValue;
    ct.stateTable[0] = 0;
    ct.stateTable[1] = 0;
    ct.symbolTT[symbolValue].deltaNbBits = 0;
    ct.symbolTT[symbolValue].deltaFindState = 0;
}

struct FseState {
    U32 value;
    U32 stateLog;
};
ZFN void fseInitState2(FseState &st, const FseCTable &ct, U32 symbol)
{
    st.stateLog = ct.tableLog;
    const FseSym tt = ct.symbolTT[symbol];
    const U32 nbBitsOut = (tt.deltaNbBits + (1u << 15)) >> 16;
    const U32 v = (nbBitsOut << 16) - tt.deltaNbBits;
    st.value = ct.stateTable[(int)(v >> nbBitsOut) + tt.deltaFindState];
}
ZFN void fseEncodeSymbol(BitW &b, FseState &st, const FseCTable &ct, U32 symbol)
{
    const FseSym tt = ct.symbolTT[symbol];
    const U32 nbBitsOut = (st.value + tt.deltaNbBits) >> 16;
    bitAdd(b, st.value, nbBitsOut);
    st.value = ct.stateTable[(int)(st.value >> nbBitsOut) + tt.deltaFindState];
}
ZFN void fseFlushState(BitW &b, const FseState &st)
{
    bitAdd(b, st.value, st.stateLog);
    bitFlush(b);
}

// ---- Huffman ----
struct HufNode {
    U32 count;
    U16 parent;
    BYTE byte, nbBits;
};
struct HufCElt {
    U16 val;
    BYTE nbBits;
};

// scratch of the entropy stage (one per frame, in the frame's workspace)
struct EntWs {
    U32 count[256];
    int16_t norm[MaxML + 2];
    U32 cumul[MaxML + 3];
    BYTE tableSymbol[512];
    HufNode huffNode0[2 * 256 + 1]; // huffNode = huffNode0 + 1
    HufCElt hufCTable[256];
    U32 rankBase[33], rankCur[33];
    BYTE huffWeight[256];
    FseCTable fse[3];     // LL, OF, ML
    FseCTable fseW;       // Huffman weights
};

ZFN U32 hufSetMaxHeight(HufNode *huffNode, U32 lastNonNull, U32 maxNbBits)
{
    const U32 largestBits = huffNode[lastNonNull].nbBits;
    if (largestBits <= maxNbBits)
        return largestBits;
    {
        int totalCost = 0;
        const U32 baseCost = 1u << (largestBits - maxNbBits);
        int n = (int)lastNonNull;
        while (huffNode[n].nbBits > maxNbBits) {
            totalCost += (int)(baseCost - (1u << (largestBits - huffNode[n].nbBits)));
            huffNode[n].nbBits = (BYTE)maxNbBits;
            n--;
        }
        while (huffNode[n].nbBits == maxNbBits)
            n--;
        totalCost >>= (largestBits - maxNbBits);
        {
            const U32 noSymbol = 0xF0F0F0F0;
            U32 rankLast[HUF_TABLELOG_MAX + 2];
            for (U32 i = 0; i < HUF_TABLELOG_MAX + 2; ++i)
                rankLast[i] = noSymbol;
            {
                U32 currentNbBits = maxNbBits;
                for (int pos = n; pos >= 0; pos--) {
                    if (huffNode[pos].nbBits >= currentNbBits)
                        continue;
                    currentNbBits = huffNode[pos].nbBits;
                    rankLast[maxNbBits - currentNbBits] = (U32)pos;
                }
            }
            while (totalCost > 0) {
                U32 nBitsToDecrease = highbit32((U32)totalCost) + 1;
                for (; nBitsToDecrease > 1; nBitsToDecrease--) {
                    const U32 highPos = rankLast[nBitsToDecrease];
                    const U32 lowPos = rankLast[nBitsToDecrease - 1];
                    if (highPos == noSymbol)
                        continue;
                    if (lowPos == noSymbol)
                        break;
                    {
                        const U32 highTotal = huffNode[highPos].count;
                        const U32 lowTotal = 2 * huffNode[lowPos].count;
                        if (highTotal <= lowTotal)
                            break;
                    }
                }
                while ((nBitsToDecrease <= HUF_TABLELOG_MAX) && (rankLast[nBitsToDecrease] == noSymbol))
                    nBitsToDecrease++;
                totalCost -= 1 << (nBitsToDecrease - 1);
                if (rankLast[nBitsToDecrease - 1] == noSymbol)
                    rankLast[nBitsToDecrease - 1] = rankLast[nBitsToDecrease];
                huffNode[rankLast[nBitsToDecrease]].nbBits++;
                if (rankLast[nBitsToDecrease] == 0)
                    rankLast[nBitsToDecrease] = noSymbol;
                else {
                    rankLast[nBitsToDecrease]--;
                    if (huffNode[rankLast[nBitsToDecrease]].nbBits != maxNbBits - nBitsToDecrease)
                        rankLast[nBitsToDecrease] = noSymbol;
                }
            }
            while (totalCost < 0) {
                if (rankLast[1] == noSymbol) {
                    while (huffNode[n].nbBits == maxNbBits)
                        n--;
                    huffNode[n + 1].nbBits--;
                    rankLast[1] = (U32)(n + 1);
                    totalCost++;
                    continue;
                }
                huffNode[rankLast[1] + 1].nbBits--;
                rankLast[1]++;
                totalCost++;
            }
        }
    }
    return maxNbBits;
}

// HUF_buildCTable_wksp; returns maxNbBits
ZFN U32 hufBuildCTable(EntWs &e, const U32 *count, U32 maxSymbolValue, U32 maxNbBits)
{
    HufNode *const huffNode0 = e.huffNode0;
    HufNode *const huffNode = huffNode0 + 1;
    const int STARTNODE = HUF_SYMBOLVALUE_MAX + 1;
    int nodeNb = STARTNODE;
    if (maxNbBits == 0)
        maxNbBits = HUF_TABLELOG_DEFAULT;
    for (U32 i = 0; i < 2 * 256 + 1; ++i) {
        huffNode0[i].count = 0;
        huffNode0[i].parent = 0;
        huffNode0[i].byte = 0;
        huffNode0[i].nbBits = 0;
    }
    // HUF_sort: decreasing count, buckets by highbit(count + 1), insertion inside a bucket
    {
        const int maxSymbolValue1 = (int)maxSymbolValue + 1;
        for (int n = 0; n < 33; ++n)
            e.rankBase[n] = e.rankCur[n] = 0;
        for (int n = 0; n < maxSymbolValue1; ++n)
            e.rankBase[highbit32(count[n] + 1)]++;
        for (int n = 31; n > 0; --n) {
            e.rankBase[n - 1] += e.rankBase[n];
            e.rankCur[n - 1] = e.rankBase[n - 1];
        }
        for (int n = 0; n < maxSymbolValue1; ++n) {
            const U32 c = count[n];
            const U32 r = highbit32(c + 1) + 1;
            U32 pos = e.rankCur[r]++;
            while ((pos > e.rankBase[r]) && (c > huffNode[pos - 1].count)) {
                huffNode[pos] = huffNode[pos - 1];
                pos--;
            }
            huffNode[pos].count = c;
            huffNode[pos].byte = (BYTE)n;
        }
    }
    int nonNullRank = (int)maxSymbolValue;
    while (huffNode[nonNullRank].count == 0)
        nonNullRank--;
    int lowS = nonNullRank;
    const int nodeRoot = nodeNb + lowS - 1;
    int lowN = nodeNb;
    huffNode[nodeNb].count = huffNode[lowS].count + huffNode[lowS - 1].count;
    huffNode[lowS].parent = huffNode[lowS - 1].parent = (U16)nodeNb;
    nodeNb++;
    lowS -= 2;
    for (int n = nodeNb; n <= nodeRoot; n++)
        huffNode[n].count = 1u << 30;
    huffNode0[0].count = 1u << 31; // fake entry, strong barrier
    while (nodeNb <= nodeRoot) {
        const int n1 = (huffNode[lowS].count < huffNode[lowN].count) ? lowS-- : lowN++;
        const int n2 = (huffNode[lowS].count < huffNode[lowN].count) ? lowS-- : lowN++;
        huffNode[nodeNb].count = huffNode[n1].count + huffNode[n2].count;
        huffNode[n1].parent = huffNode[n2].parent = (U16)nodeNb;
        nodeNb++;
    }
    huffNode[nodeRoot].nbBits = 0;
    for (int n = nodeRoot - 1; n >= STARTNODE; n--)
        huffNode[n].nbBits = (BYTE)(huffNode[huffNode[n].parent].nbBits + 1);
    for (int n = 0; n <= nonNullRank; n++)
        huffNode[n].nbBits = (BYTE)(huffNode[huffNode[n].parent].nbBits + 1);
    maxNbBits = hufSetMaxHeight(huffNode, (U32)nonNullRank, maxNbBits);
    {
        U16 nbPerRank[HUF_TABLELOG_MAX + 1], valPerRank[HUF_TABLELOG_MAX + 1];
        for (U32 i = 0; i <= HUF_TABLELOG_MAX; ++i)
            nbPerRank[i] = valPerRank[i] = 0;
        const int alphabetSize = (int)(maxSymbolValue + 1);
        for (int n = 0; n <= nonNullRank; n++)
            nbPerRank[huffNode[n].nbBits]++;
        {
            U16 min = 0;
            for (int n = (int)maxNbBits; n > 0; n--) {
                valPerRank[n] = min;
                min = (U16)(min + nbPerRank[n]);
                min >>= 1;
            }
        }
        for (int n = 0; n < alphabetSize; n++)
            e.hufCTable[huffNode[n].byte].nbBits = huffNode[n].nbBits;
        for (int n = 0; n < alphabetSize; n++)
            e.hufCTable[n].val = valPerRank[e.hufCTable[n].nbBits]++;
    }
    return maxNbBits;
}

// HUF_compressWeights: 0 = not compressible, 1 = rle, else size
ZFN U32 hufCompressWeights(EntWs &e, BYTE *dst, const BYTE *weightTable, U32 wtSize)
{
    BYTE *op = dst;
    U32 maxSymbolValue = HUF_TABLELOG_MAX;
    U32 tableLog = 6; // MAX_FSE_TABLELOG_FOR_HUFF_HEADER
    U32 *count = e.count;
    if (wtSize <= 1)
        return 0;
    {
        for (U32 s = 0; s <= maxSymbolValue; ++s)
            count[s] = 0;
        for (U32 i = 0; i < wtSize; ++i)
            count[weightTable[i]]++;
        while (!count[maxSymbolValue])
            maxSymbolValue--;
        U32 maxCount = 0;
        for (U32 s = 0; s <= maxSymbolValue; ++s)
            if (count[s] > maxCount)
                maxCount = count[s];
        if (maxCount == wtSize)
            return 1;
        if (maxCount == 1)
            return 0;
    }
    tableLog = fseOptimalTableLog(tableLog, wtSize, maxSymbolValue, 2);
    if (fseNormalizeCount(e.norm, tableLog, count, wtSize, maxSymbolValue, false) == ~0u)
        return 0; // (the library forwards an error here; it cannot happen for valid weights)
    {
        const U32 hSize = fseWriteNCount(op, e.norm, maxSymbolValue, tableLog);
        op += hSize;
    }
    fseBuildCTable(e.fseW, e.norm, maxSymbolValue, tableLog, e.tableSymbol, e.cumul);
    // FSE_compress_usingCTable: two interleaved states, symbols from the end
    {
        if (wtSize <= 2)
            return 0;
        const BYTE *ip = weightTable + wtSize;
        BitW b;
        bitInit(b, op);
        FseState s1, s2;
        U32 rest = wtSize;
        if (rest & 1) {
            fseInitState2(s1, e.fseW, *--ip);
            fseInitState2(s2, e.fseW, *--ip);
            fseEncodeSymbol(b, s1, e.fseW, *--ip);
            bitFlush(b);
        } else {
            fseInitState2(s2, e.fseW, *--ip);
            fseInitState2(s1, e.fseW, *--ip);
        }
        while (ip > weightTable) {
            fseEncodeSymbol(b, s2, e.fseW, *--ip);
            fseEncodeSymbol(b, s1, e.fseW, *--ip);
            bitFlush(b);
        }
        fseFlushState(b, s2);
        fseFlushState(b, s1);
        const U32 cSize = bitClose(b);
        op += cSize;
    }
    return (U32)(op - dst);
}

// HUF_writeCTable; returns the header size, 0 on error
ZFN U32 hufWriteCTable(EntWs &e, BYTE *dst, U32 maxSymbolValue, U32 huffLog)
{
    BYTE bitsToWeight[HUF_TABLELOG_MAX + 1];
    BYTE *const huffWeight = e.huffWeight;
    BYTE *op = dst;
    bitsToWeight[0] = 0;
    for (U32 n = 1; n < huffLog + 1; n++)
        bitsToWeight[n] = (BYTE)(huffLog + 1 - n);
    for (U32 n = 0; n < maxSymbolValue; n++)
        huffWeight[n] = bitsToWeight[e.hufCTable[n].nbBits];
    {
        const U32 hSize = hufCompressWeights(e, op + 1, huffWeight, maxSymbolValue);
        if ((hSize > 1) & (hSize < maxSymbolValue / 2)) {
            op[0] = (BYTE)hSize;
            return hSize + 1;
        }
    }
    if (maxSymbolValue > (256 - 128))
        return 0;
    op[0] = (BYTE)(128 + (maxSymbolValue - 1));
    huffWeight[maxSymbolValue] = 0;
    for (U32 n = 0; n < maxSymbolValue; n += 2)
        op[(n / 2) + 1] = (BYTE)((huffWeight[n] << 4) + huffWeight[n + 1]);
    return ((maxSymbolValue + 1) / 2) + 1;
}

// HUF_compress1X_usingCTable: symbols from the end
// `limit`: a stream that reaches it is longer than anything the caller would keep (HUF_compress's own capacity check answers 0 =
// "not compressible" there: the literals then go out raw either way); nothing is written beyond limit + 8
ZFN U32 hufCompress1X(const EntWs &e, BYTE *dst, const BYTE *src, U32 srcSize, const BYTE *limit)
{
    BitW b;
    bitInit(b, dst);
    for (U32 n = srcSize; n > 0; --n) {
        const HufCElt c = e.hufCTable[src[n - 1]];
        bitAdd(b, c.val, c.nbBits);
        if (b.ptr >= limit)
            return 0;
    }
    return bitClose(b);
}

// HUF_compress1X_repeat / HUF_compress4X_repeat without a previous table: 0 = not compressible, 1 = rle (byte in dst[0])
ZFN U32 hufCompress(EntWs &e, BYTE *dst, const BYTE *src, U32 srcSize, bool singleStream)
{
    BYTE *op = dst;
    U32 maxSymbolValue = HUF_SYMBOLVALUE_MAX;
    U32 huffLog = HUF_TABLELOG_DEFAULT;
    U32 *count = e.count;
    if (!srcSize)
        return 0;
    {
        for (U32 s = 0; s < 256; ++s)
            count[s] = 0;
        for (U32 i = 0; i < srcSize; ++i)
            count[src[i]]++;
        while (!count[maxSymbolValue])
            maxSymbolValue--;
        U32 largest = 0;
        for (U32 s = 0; s <= maxSymbolValue; ++s)
            if (count[s] > largest)
                largest = count[s];
        if (largest == srcSize) {
            *dst = src[0];
            return 1;
        }
        if (largest <= (srcSize >> 7) + 4)
            return 0;
    }
    huffLog = fseOptimalTableLog(huffLog, srcSize, maxSymbolValue, 1);
    huffLog = hufBuildCTable(e, count, maxSymbolValue, huffLog); // (e.count is free afterwards: the weight compression reuses it)
    for (U32 s = maxSymbolValue + 1; s < 256; ++s) {
        e.hufCTable[s].val = 0;
        e.hufCTable[s].nbBits = 0;
    }
    {
        const U32 hSize = hufWriteCTable(e, op, maxSymbolValue, huffLog);
        if (!hSize)
            return 0;
        if (hSize + 12u >= srcSize)
            return 0;
        op += hSize;
    }
    // (a result of srcSize - 1 bytes or more is dropped below: a stream that gets there is cut short)
    const BYTE *const limit = dst + (srcSize > 1 ? srcSize - 1 : 0);
    if (singleStream) {
        const U32 c = hufCompress1X(e, op, src, srcSize, limit);
        if (!c)
            return 0;
        op += c;
    } else {
        const U32 segmentSize = (srcSize + 3) / 4;
        const BYTE *ip = src;
        BYTE *const jump = op;
        if (srcSize < 12)
            return 0;
        op += 6;
        for (int sidx = 0; sidx < 3; ++sidx) {
            const U32 c = hufCompress1X(e, op, ip, segmentSize, limit);
            if (!c)
                return 0;
            jump[2 * sidx] = (BYTE)c;
            jump[2 * sidx + 1] = (BYTE)(c >> 8);
            op += c;
            ip += segmentSize;
        }
        const U32 c = hufCompress1X(e, op, ip, (U32)(src + srcSize - ip), limit);
        if (!c)
            return 0;
        op += c;
    }
    if ((U32)(op - dst) >= srcSize - 1)
        return 0;
    return (U32)(op - dst);
}

ZFN U32 minGain(U32 srcSize, U32 strat)
{
    const U32 minlog = (strat >= STRAT_BTULTRA) ? strat - 1 : 6;
    return (srcSize >> minlog) + 2;
}

ZFN U32 noCompressLiterals(BYTE *ostart, const BYTE *src, U32 srcSize, U32 type, U32 payload)
{
    const U32 flSize = 1 + (srcSize > 31) + (srcSize > 4095);
    switch (flSize) {
    case 1: ostart[0] = (BYTE)(type + (srcSize << 3)); break;
    case 2: {
        const U32 v = type + (1 << 2) + (srcSize << 4);
        ostart[0] = (BYTE)v;
        ostart[1] = (BYTE)(v >> 8);
        break;
    }
    default: {
        const U32 v = type + (3 << 2) + (srcSize << 4);
        ostart[0] = (BYTE)v;
        ostart[1] = (BYTE)(v >> 8);
        ostart[2] = (BYTE)(v >> 16);
        ostart[3] = (BYTE)(v >> 24); // MEM_writeLE32 of a 3-byte header: the 4th byte is overwritten by the payload
        break;
    }
    }
    for (U32 i = 0; i < payload; ++i)
        ostart[flSize + i] = src[i];
    return flSize + payload;
}

// ZSTD_compressLiterals, first block (no previous Huffman table)
ZFN U32 compressLiterals(EntWs &e, U32 strategy, BYTE *ostart, const BYTE *src, U32 srcSize)
{
    const U32 mg = minGain(srcSize, strategy);
    const U32 lhSize = 3 + (srcSize >= 1024) + (srcSize >= 16384);
    const bool singleStream = srcSize < 256;
    if (srcSize <= 63) // COMPRESS_LITERALS_SIZE_MIN
        return noCompressLiterals(ostart, src, srcSize, set_basic, srcSize);
    const U32 cLitSize = hufCompress(e, ostart + lhSize, src, srcSize, singleStream);
    if ((cLitSize == 0) | (cLitSize >= srcSize - mg))
        return noCompressLiterals(ostart, src, srcSize, set_basic, srcSize);
    if (cLitSize == 1)
        return noCompressLiterals(ostart, src, srcSize, set_rle, 1);
    const U32 hType = set_compressed;
    switch (lhSize) {
    case 3: {
        const U32 lhc = hType + ((U32)(!singleStream) << 2) + (srcSize << 4) + (cLitSize << 14);
        ostart[0] = (BYTE)lhc;
        ostart[1] = (BYTE)(lhc >> 8);
        ostart[2] = (BYTE)(lhc >> 16);
        break;
    }
    case 4: {
        const U32 lhc = hType + (2 << 2) + (srcSize << 4) + (cLitSize << 18);
        ostart[0] = (BYTE)lhc;
        ostart[1] = (BYTE)(lhc >> 8);
        ostart[2] = (BYTE)(lhc >> 16);
        ostart[3] = (BYTE)(lhc >> 24);
        break;
    }
    default: {
        const U32 lhc = hType + (3 << 2) + (srcSize << 4) + (cLitSize << 22);
        ostart[0] = (BYTE)lhc;
        ostart[1] = (BYTE)(lhc >> 8);
        ostart[2] = (BYTE)(lhc >> 16);
        ostart[3] = (BYTE)(lhc >> 24);
        ostart[4] = (BYTE)(cLitSize >> 10);
        break;
    }
    }
    return lhSize + cLitSize;
}

// ---- sequence tables ----
ZFN U32 crossEntropyCost(const int16_t *norm, U32 accuracyLog, const U32 *count, U32 max)
{
    const U32 shift = 8 - accuracyLog;
    U64 cost = 0;
    for (U32 s = 0; s <= max; ++s) {
        const U32 normAcc = (norm[s] != -1) ? (U32)norm[s] : 1;
        const U32 norm256 = normAcc << shift;
        cost += (U64)count[s] * kInverseProbabilityLog256[norm256];
    }
    return (U32)(cost >> 8);
}

ZFN U32 entropyCost(const U32 *count, U32 max, U32 total)
{
    U32 cost = 0;
    for (U32 s = 0; s <= max; ++s) {
        U32 norm = (256 * count[s]) / total;
        if (count[s] != 0 && norm == 0)
            norm = 1;
        cost += count[s] * kInverseProbabilityLog256[norm];
    }
    return cost >> 8;
}

// ZSTD_selectEncodingType for strategy >= lazy, first block (no table to repeat)
ZFN U32 selectEncodingType(EntWs &e, const U32 *count, U32 max, U32 mostFrequent, U32 nbSeq, U32 FSELog, const int16_t *defaultNorm,
                           U32 defaultNormLog, bool isDefaultAllowed, BYTE *scratch)
{
    if (mostFrequent == nbSeq) {
        if (isDefaultAllowed && nbSeq <= 2)
            return set_basic;
        return set_rle;
    }
    const U64 HUGE = ~0ULL;
    const U64 basicCost = isDefaultAllowed ? crossEntropyCost(defaultNorm, defaultNormLog, count, max) : HUGE;
    // ZSTD_NCountCost
    U64 NCountCost;
    {
        const U32 tableLog = fseOptimalTableLog(FSELog, nbSeq, max, 2);
        fseNormalizeCount(e.norm, tableLog, count, nbSeq, max, nbSeq >= 2048);
        NCountCost = fseWriteNCount(scratch, e.norm, max, tableLog);
    }
    const U64 compressedCost = (NCountCost << 3) + entropyCost(count, max, nbSeq);
    if (basicCost <= compressedCost) // (repeatCost is an error code here: larger than everything)
        return set_basic;
    return set_compressed;
}

// ZSTD_buildCTable; returns the bytes written at op
ZFN U32 buildSeqCTable(EntWs &e, BYTE *op, FseCTable &ct, U32 FSELog, U32 type, U32 *count, U32 max, const BYTE *codeTable, U32 nbSeq,
                       const int16_t *defaultNorm, U32 defaultNormLog, U32 defaultMax)
{
    switch (type) {
    case set_rle:
        fseBuildCTableRle(ct, max);
        *op = codeTable[0];
        return 1;
    case set_basic:
        fseBuildCTable(ct, defaultNorm, defaultMax, defaultNormLog, e.tableSymbol, e.cumul);
        return 0;
    default: {
        U32 nbSeq_1 = nbSeq;
        const U32 tableLog = fseOptimalTableLog(FSELog, nbSeq, max, 2);
        if (count[codeTable[nbSeq - 1]] > 1) {
            count[codeTable[nbSeq - 1]]--;
            nbSeq_1--;
        }
        fseNormalizeCount(e.norm, tableLog, count, nbSeq_1, max, nbSeq_1 >= 2048);
        const U32 NCountSize = fseWriteNCount(op, e.norm, max, tableLog);
        fseBuildCTable(ct, e.norm, max, tableLog, e.tableSymbol, e.cumul);
        return NCountSize;
    }
    }
}

ZFN U32 histCodes(U32 *count, U32 *maxPtr, const BYTE *codes, U32 n)
{
    U32 max = *maxPtr;
    for (U32 s = 0; s <= max; ++s)
        count[s] = 0;
    for (U32 i = 0; i < n; ++i)
        count[codes[i]]++;
    while (!count[max])
        max--;
    *maxPtr = max;
    U32 largest = 0;
    for (U32 s = 0; s <= max; ++s)
        if (count[s] > largest)
            largest = count[s];
    return largest;
}

// ZSTD_entropyCompressSequences_internal + the compressibility check of ZSTD_entropyCompressSequences.
// seqs / lits: the block's sequence store (lits includes the last literals); codes: 3 * nSeq bytes of scratch.
// returns the compressed block size, 0 = emit a raw block
ZFN U32 entropyCompressBlock(EntWs &e, const CParams &cp, const Seq *seqs, U32 nbSeq, const BYTE *lits, U32 litSize, BYTE *codes, BYTE *dst,
                             U32 srcSize)
{
    BYTE *op = dst;
    BYTE *const llCodeTable = codes, *const ofCodeTable = codes + nbSeq, *const mlCodeTable = codes + 2 * (size_t)nbSeq;
    op += compressLiterals(e, cp.strategy, op, lits, litSize);
    if (nbSeq < 128)
        *op++ = (BYTE)nbSeq;
    else if (nbSeq < LONGNBSEQ) {
        op[0] = (BYTE)((nbSeq >> 8) + 0x80);
        op[1] = (BYTE)nbSeq;
        op += 2;
    } else {
        op[0] = 0xFF;
        op[1] = (BYTE)(nbSeq - LONGNBSEQ);
        op[2] = (BYTE)((nbSeq - LONGNBSEQ) >> 8);
        op += 3;
    }
    if (nbSeq != 0) {
        BYTE *const seqHead = op++;
        BYTE *lastNCount = nullptr;
        for (U32 u = 0; u < nbSeq; u++) { // ZSTD_seqToCodes
            llCodeTable[u] = (BYTE)LLcode(seqs[u].litLength);
            ofCodeTable[u] = (BYTE)highbit32(seqs[u].offCode + 1);
            mlCodeTable[u] = (BYTE)MLcode(seqs[u].matchLength - MINMATCH);
        }
        U32 *const count = e.count;
        U32 LLtype, Offtype, MLtype;
        {
            U32 max = MaxLL;
            const U32 mostFrequent = histCodes(count, &max, llCodeTable, nbSeq);
            LLtype = selectEncodingType(e, count, max, mostFrequent, nbSeq, LLFSELog, LL_defaultNorm, LL_defaultNormLog, true, op);
            const U32 countSize = buildSeqCTable(e, op, e.fse[0], LLFSELog, LLtype, count, max, llCodeTable, nbSeq, LL_defaultNorm, LL_defaultNormLog, MaxLL);
            if (LLtype == set_compressed)
                lastNCount = op;
            op += countSize;
        }
        {
            U32 max = MaxOff;
            const U32 mostFrequent = histCodes(count, &max, ofCodeTable, nbSeq);
            const bool defaultAllowed = max <= DefaultMaxOff;
            Offtype = selectEncodingType(e, count, max, mostFrequent, nbSeq, OffFSELog, OF_defaultNorm, OF_defaultNormLog, defaultAllowed, op);
            const U32 countSize = buildSeqCTable(e, op, e.fse[1], OffFSELog, Offtype, count, max, ofCodeTable, nbSeq, OF_defaultNorm, OF_defaultNormLog, DefaultMaxOff);
            if (Offtype == set_compressed)
                lastNCount = op;
            op += countSize;
        }
        {
            U32 max = MaxML;
            const U32 mostFrequent = histCodes(count, &max, mlCodeTable, nbSeq);
            MLtype = selectEncodingType(e, count, max, mostFrequent, nbSeq, MLFSELog, ML_defaultNorm, ML_defaultNormLog, true, op);
            const U32 countSize = buildSeqCTable(e, op, e.fse[2], MLFSELog, MLtype, count, max, mlCodeTable, nbSeq, ML_defaultNorm, ML_defaultNormLog, MaxML);
            if (MLtype == set_compressed)
                lastNCount = op;
            op += countSize;
        }
        *seqHead = (BYTE)((LLtype << 6) + (Offtype << 4) + (MLtype << 2));
        { // ZSTD_encodeSequences
            BitW b;
            bitInit(b, op);
            FseState stML, stOF, stLL;
            const FseCTable &ctLL = e.fse[0], &ctOF = e.fse[1], &ctML = e.fse[2];
            fseInitState2(stML, ctML, mlCodeTable[nbSeq - 1]);
            fseInitState2(stOF, ctOF, ofCodeTable[nbSeq - 1]);
            fseInitState2(stLL, ctLL, llCodeTable[nbSeq - 1]);
            bitAdd(b, seqs[nbSeq - 1].litLength, LLbits(llCodeTable[nbSeq - 1]));
            bitAdd(b, seqs[nbSeq - 1].matchLength - MINMATCH, MLbits(mlCodeTable[nbSeq - 1]));
            bitAdd(b, seqs[nbSeq - 1].offCode + 1, ofCodeTable[nbSeq - 1]);
            bitFlush(b);
            for (U32 n = nbSeq - 2; n < nbSeq; n--) { // intentional underflow
                const BYTE llCode = llCodeTable[n], ofCode = ofCodeTable[n], mlCode = mlCodeTable[n];
                fseEncodeSymbol(b, stOF, ctOF, ofCode);
                fseEncodeSymbol(b, stML, ctML, mlCode);
                fseEncodeSymbol(b, stLL, ctLL, llCode);
                bitFlush(b);
                bitAdd(b, seqs[n].litLength, LLbits(llCode));
                bitAdd(b, seqs[n].matchLength - MINMATCH, MLbits(mlCode));
                bitFlush(b);
                bitAdd(b, seqs[n].offCode + 1, ofCode);
                bitFlush(b);
            }
            fseFlushState(b, stML);
            fseFlushState(b, stOF);
            fseFlushState(b, stLL);
            op += bitClose(b);
            if (lastNCount && (op - lastNCount) < 4)
                return 0; // decoder bug of zstd <= 1.3.4: the library emits a raw block instead
        }
    }
    const U32 cSize = (U32)(op - dst);
    if (cSize >= srcSize - minGain(srcSize, cp.strategy))
        return 0;
    return cSize;
}

} // namespace zs
