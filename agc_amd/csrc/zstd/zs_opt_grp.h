// Derived from Zstandard 1.4.9 (Copyright (c) 2016-present, Facebook, Inc.; BSD license): see NOTICE in this directory.
// zs_opt_grp.h -- the optimal parser of zs_opt.h / zs_opt_sm.h with SEVERAL LANES PER FRAME (a "group" of G lanes).
//
// Why: one lane per frame (zs_opt_sm.h) leaves a frame with ONE dependent chain of ~11 memory round trips per position, and a
// Close() has only as many chains as it has packs: the launch lasts as long as 2 x 13 000 positions x 11 HBM latencies.  Here
// the G lanes of a group take G CONSECUTIVE positions of the forward pass in the same trip of the micro-step loop:
//   * with minMatch = 3 the price-table entries of cur, cur+1 and cur+2 are final once the literal step has been chained
//     through them (a match found at cur lands on cur+3 at the earliest), so their repcodes, hash-3 probes, tree walks and
//     price updates run side by side and their memory waits coincide;
//   * a lane's tree walk is READ-ONLY: the stores of ZSTD_insertBtAndGetAllMatches are recorded and replayed ("commit") after
//     the group has been validated IN ORDER -- positions of different hash buckets live in disjoint trees, so a walk that
//     started before the commit of the lanes in front of it has seen exactly what the sequential code would have seen; equal
//     buckets, skipped areas, early exits, long walks ... ("anomalies") cut the group at that lane, whose position is then
//     redone by the plain one-lane path of zs_opt_sm.h (kept below, state for state);
//   * the price updates of the lanes are merged per target position in lane order with the library's strict `<`, so ties go
//     to the earlier position exactly as in the sequential loop.
// The result is the library's parse, decision for decision (tests/test_zstd_frames.py runs this file on the host, one loop
// iteration per lane and segment, against libzstd 1.4.9; tests/test_gpu_zstd.py runs the kernel).
//
// Code shape: the trip body is a sequence of SEGMENTS; lanes talk to each other only through the group's exchange record
// (GrpX, LDS on the device) written in one segment and read in a later one.  On the device every lane runs every segment once
// per trip (same-wave LDS accesses are ordered); the host build runs each segment for lane 0..G-1 in turn.
#pragma once
#include "zs_opt.h"
#include "zs_opt_sm.h"

namespace zs {

#ifdef ZS_GRP_STATS
static unsigned long long g_grp_trips, g_grp_valid[4], g_grp_plan[4];
#endif
constexpr U32 GRP_MAX = 3;   // lanes per group the exchange record is sized for (= minMatch of the bt* levels <= 17)
constexpr U32 GRP_MC = 6;    // matches of a request a lane keeps in registers (99.9 % of the requests have <= 5)
constexpr U32 GRP_RC = 16;   // tree stores a walk may record: 14 levels + the two closing zeros (99.5 % of the walks)
constexpr U32 GRP_PT = 8;    // price targets per trip
#ifndef ZS_GRP_STORE_SEQS
#define ZS_GRP_STORE_SEQS 4
#endif
constexpr U32 GRP_STORE_SEQS = ZS_GRP_STORE_SEQS; // sequences of a finished chunk stored per trip (<= 4: one 8-byte read of the path list)
constexpr U32 GRP_ST_LITS = 4;    // ... without a dependent chain when their literal runs are at most this long (99 % of the runs)
constexpr U32 GRP_ST_LAST = 0xFFFFFFF0u, GRP_ST_DONE = 0xFFFFFFF1u; // store cursor: lastSequence is next / the chunk is stored
#ifndef ZS_GRP_WALK_LEVELS
#define ZS_GRP_WALK_LEVELS 12
#endif
constexpr U32 GRP_WALK_LEVELS = ZS_GRP_WALK_LEVELS; // common tree levels per trip (a walk has 3.7 on average, 98 % have <= 12)

// (a pointer type that SAYS it points into LDS: the compiler must not merge a recorded store with the tree store it replaces
// into one store through a generic pointer)
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) U32 *ZS_LDS_U32P;
#else
typedef U32 *ZS_LDS_U32P;
#endif

enum { ST_F_IDLE = 32, ST_G_BEGIN, ST_G_WAIT, ST_G_PRICE, ST_G_FIRST };
enum { GS_OK = 0, GS_ANOMALY = 1 };

// what the lanes of one group tell each other (one record per group; device: LDS)
struct GrpX {
    U32 g;             // bits 0-7: lanes taking part in the trip that starts now (0: none); bit 8 ("plain"): the leader's position can
                       // go the group way (no skipped positions to insert first)
    U32 done;          // the leader has finished the block
    U32 cur, last_pos, ip;
    U32 pend_n, pend_h3;    // the position the leader has just put into the hash-3 table (q0 - 1, if pend_n), and its hash
    U32 priceType, litSumBP, llSumBP, mlSumBP, ocSumBP; // the price model's bases (change when a chunk has been stored)
    int oc_price;      // literal chain: the entry of the lane in front
    U32 oc_mlen, oc_litlen, oc_rep[3];
    U32 h[GRP_MAX], h3[GRP_MAX];
    U32 wdone[GRP_MAX]; // 0: the lane's walk is under way, 1: finished, 2: finished as an anomaly (GS_ANOMALY + 1)
    U32 nbm[GRP_MAX], maxML[GRP_MAX], maxOff[GRP_MAX], mEnd[GRP_MAX], qlit[GRP_MAX], litback[GRP_MAX];
    int cand[GRP_MAX][GRP_PT];
};
constexpr U32 GRPX_WORDS = (sizeof(GrpX) / 4) | 1; // odd stride in LDS: equal fields of neighbouring groups on different banks

// everything one lane carries through the loop (device: registers)
struct GLane {
    OptWs w;           // the lane's own copy: pointers and constants are equal in the group, the counters are the leader's
    U32 j;             // lane of the group, 0 = leader
    U32 state;
    ZS_LDS_U32P recs;  // GRP_RC recorded tree stores (device: LDS)
    bool wide;         // ... of two words each (slot, value) instead of one packed word: inputs beyond the btultra2 class
    // parser state (leader)
    U32 ip, anchor, cur, last_pos, adv;
    Optimal lastSequence;
    bool inChunk;
    U32 rep0, rep1, rep2; // the block's repcodes
    U32 nextToUpdate3;
    // match request
    U32 q_current, q_ll0, q_litlen, q_rep0, q_rep1, q_rep2, basePrice, nbMatches, cur_litlen_back;
    // tree walk
    U32 wk_current, matchIndex, clSmaller, clLarger, smallerPtr, largerPtr, matchEndIdx, bestLength, nbCompares, btLow, lowLimit, mnum, upd_idx;
    bool rec;          // the walk records its stores instead of making them
    bool grp;          // the walk belongs to a group trip (its end is reported to the group)
    // the next level's node, read ahead (grpWalkIssue): both children, 8 bytes of the match and of the position at the length
    // the two are already known to share
    bool wk_pre;
    U64 wk_pair, wk_mb, wk_pb;
    U32 nrec, gstatus;
    U32 m_off[GRP_MC], m_len[GRP_MC]; // the first matches of the request
    U32 last_m_off, last_m_len;
    // price loops of the one-lane path
    U32 pr_matchNb, pr_pos, pr_literalsPrice, pm_off, pm_len, pm_start;
    // store loop
    U32 storePos, storeEnd;
    U64 p8;            // the 8 source bytes at the position being worked on
    U32 p8_pos;
    // group trip
    U32 g, g_cur, g_v, g_lp0, tcur, t1;
    Optimal oc, op;
    U32 pr0, pr1, pr2; // repcodes of the entry oc came from (read ahead)
    U32 lit_freq, h, h3, mi0, mi3;
    int cand[GRP_PT];  // the lane's prices for the targets of this trip
    U32 coff[GRP_PT];  // ... and the offset codes they belong to
    int oldp[GRP_PT];  // the prices the first targets hold before this trip (read in segment B)
    U32 oldp_t0;       // ... and the first of those targets
};

// a recorded store in one word: slot < 2^15 (chainLog <= 15), value < 2^17 (index <= 2 * 16 KiB + 1) -- the btultra2 class
ZFN U32 grpRecPack(U32 slot, U32 val) { return (val << 15) | slot; }
ZFN void grpRecPut(GLane &l, U32 slot, U32 val)
{
    if (l.wide) {
        l.recs[2 * l.nrec] = slot;
        l.recs[2 * l.nrec + 1] = val;
    } else
        l.recs[l.nrec] = grpRecPack(slot, val);
    l.nrec++;
}
ZFN void grpRecGet(const GLane &l, U32 r, U32 &slot, U32 &val)
{
    if (l.wide) {
        slot = l.recs[2 * r];
        val = l.recs[2 * r + 1];
    } else {
        const U32 x = l.recs[r];
        slot = x & 0x7FFFu;
        val = x >> 15;
    }
}

ZFN U32 grpSel3(U32 a0, U32 a1, U32 a2, U32 i) { return i == 0 ? a0 : (i == 1 ? a1 : a2); }

ZFN void grpMSet(GLane &l, U32 i, U32 off, U32 len)
{
    if (i < GRP_MC) {
#pragma unroll
        for (U32 m = 0; m < GRP_MC; ++m)
            if (m == i) {
                l.m_off[m] = off;
                l.m_len[m] = len;
            }
    } else {
        l.w.matches[i].off = off;
        l.w.matches[i].len = len;
    }
}
ZFN void grpMGet(const GLane &l, U32 i, U32 &off, U32 &len)
{
    if (i < GRP_MC) {
        U32 o = 0, n = 0;
#pragma unroll
        for (U32 m = 0; m < GRP_MC; ++m)
            if (m == i) {
                o = l.m_off[m];
                n = l.m_len[m];
            }
        off = o;
        len = n;
    } else {
        off = l.w.matches[i].off;
        len = l.w.matches[i].len;
    }
}


// (larger inputs -- up to one block: tree slots < 2^18, indices < 2^18 -- take two words per record: GLane::wide)
ZHD bool grpEligible(const CParams &cp, U32 srcSize)
{
    return cp.minMatch == 3 && cp.strategy >= STRAT_BTULTRA && srcSize >= 8 && srcSize <= BLOCKSIZE_MAX && cp.chainLog <= 18;
}
// the packed one-word record holds the btultra2 class (inputs <= 16 KiB); everything else takes two words
ZHD bool grpWide(const CParams &cp, U32 srcSize) { return !(cp.chainLog <= 15 && srcSize <= (1u << 14)); }

ZFN void grpPublishBases(GrpX &sh, const OptWs &w)
{
    sh.priceType = w.priceType;
    sh.litSumBP = w.litSumBasePrice;
    sh.llSumBP = w.litLengthSumBasePrice;
    sh.mlSumBP = w.matchLengthSumBasePrice;
    sh.ocSumBP = w.offCodeSumBasePrice;
}

// A segment ends with ZS_GRP_END.  On the device that is a CONVERGENCE point: the lanes of a wave run in lockstep, but the
// compiler may thread one lane's path from a block of segment A straight into its block of segment B (it knows the state the
// lane leaves A with) -- the two sides of that divergent branch then run one after the other, and a follower could read the
// exchange record before the leader's side has written it (observed: a hang, the leader waiting for a follower that never
// joined).  __builtin_amdgcn_wave_barrier() is a convergent operation (no instruction): it cannot be duplicated into or moved
// across divergent paths, so every lane passes the end of segment A before any lane starts segment B; the wavefront-scope
// fences keep the compiler from moving LDS / global accesses across it.
// (ZS_GRP_PROF: a variant build for scripts/ probes -- cycles per segment of the trip, summed per wave: zs_prof[k])
#if defined(__HIP_DEVICE_COMPILE__) && defined(ZS_GRP_PROF)
#define ZS_GRP_TICK(k)                                              \
    do {                                                            \
        const unsigned long long t_ = __builtin_readcyclecounter(); \
        zs_prof[k] += t_ - zs_prof_t;                               \
        zs_prof_t = t_;                                             \
    } while (0)
#else
#define ZS_GRP_TICK(k)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define ZS_GRP_EACH(l) { GLane &l = lanes[0];
#define ZS_GRP_END                                              \
    }                                                           \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      \
    __builtin_amdgcn_wave_barrier();                            \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#define ZS_OPAQUE(x) asm volatile("" : "+v"(x))
#else
#define ZS_GRP_EACH(l) for (int li_ = 0; li_ < G; ++li_) { GLane &l = lanes[li_];
#define ZS_GRP_END }
#define ZS_OPAQUE(x) asm volatile("" : "+r"(x))
#endif

// The record of a walk is full.  The LEADER's position is the next one in the sequential order whatever the lanes behind it
// find: its stores can be made now (the lanes behind walk other trees -- one that shares the leader's bucket is an anomaly
// already) and the walk goes on storing directly.  A follower's walk is given up (false): the position will be a leader's.
ZFN bool grpRecFull(GLane &l, U32 *bt)
{
    if (l.j != 0)
        return false;
    for (U32 r = 0; r < l.nrec; ++r) {
        U32 slot, val;
        grpRecGet(l, r, slot, val);
        bt[slot] = val;
    }
    l.nrec = 0;
    l.rec = false;
    return true;
}

// reads ahead what the next level of the walk will look at (nothing, if the walk ends there): the waits of the walk's first level
// coincide with those of the repcode tests, and a level's wait with the work that follows the level before
ZFN void grpWalkIssue(GLane &l, const BYTE *src, const BYTE *iend, const U32 *bt, U32 btMask)
{
    l.wk_pre = false;
    if (l.nbCompares && (l.matchIndex >= l.lowLimit)) {
        const U32 ml0 = l.clSmaller < l.clLarger ? l.clSmaller : l.clLarger;
        const BYTE *const p = src + (l.wk_current - l.w.idx0) + ml0;
        if (p + 8 <= iend) {
            l.wk_pair = *(const U64 *)(bt + 2 * (l.matchIndex & btMask));
            l.wk_mb = read64(src + (l.matchIndex - l.w.idx0) + ml0);
            // (the position's own bytes: in a register for the first level; a second 8-byte window kept for the deeper levels
            // was spilled by the register allocator and cost a scratch round trip per level -- measured slower than this load)
            if (ml0 == 0 && l.p8_pos == l.wk_current - l.w.idx0)
                l.wk_pb = l.p8;
            else
                l.wk_pb = read64(p);
            l.wk_pre = true;
        }
    }
}

// one level of a binary-tree walk (ZSTD_insertBt1 when `upd`, else ZSTD_insertBtAndGetAllMatches); returns true when the
// walk has ended (the closing stores made or recorded).  `abort` = the record is full: the walk is given up (group lanes only).
template <bool UPD> ZFN bool grpWalkLevel(GLane &l, const BYTE *src, const BYTE *iend, U32 *bt, U32 btMask, bool &abort)
{
    bool ended = true;
    abort = false;
    if (l.nbCompares && (l.matchIndex >= l.lowLimit)) {
        if (!UPD && l.rec && l.nrec + 3 > GRP_RC && !grpRecFull(l, bt)) {
            abort = true;
            return true;
        }
        l.nbCompares--;
        const BYTE *const p = src + (l.wk_current - l.w.idx0);
        const U32 nextPtr = 2 * (l.matchIndex & btMask);
        U32 matchLength = l.clSmaller < l.clLarger ? l.clSmaller : l.clLarger;
        const BYTE *const match = src + (l.matchIndex - l.w.idx0);
        // both children are read before the stores below (a store to this walk's own pointers never hits the node read here)
        U32 childSmaller, childLarger;
        U32 pByte = 0, mByte = 0;
        bool differ = false;
        if (!UPD && l.wk_pre) {
            childLarger = (U32)l.wk_pair;          // nextPtr[0]
            childSmaller = (U32)(l.wk_pair >> 32); // nextPtr[1]
            const U64 d = l.wk_pb ^ l.wk_mb;
            if (d) {
                const U32 sh = (U32)__builtin_ctzll(d) & ~7u;
                pByte = (U32)(l.wk_pb >> sh) & 0xFF;
                mByte = (U32)(l.wk_mb >> sh) & 0xFF;
                differ = true;
                matchLength += sh >> 3;
            } else {
                matchLength += 8;
                matchLength += countEx(p + matchLength, match + matchLength, iend, &pByte, &mByte, &differ);
            }
        } else {
            const U64 pair = *(const U64 *)(bt + nextPtr);
            childLarger = (U32)pair;
            childSmaller = (U32)(pair >> 32);
            matchLength += countEx(p + matchLength, match + matchLength, iend, &pByte, &mByte, &differ);
        }
        bool brk = false;
        if (UPD) {
            if (matchLength > l.bestLength) {
                l.bestLength = matchLength;
                if (matchLength > l.matchEndIdx - l.matchIndex)
                    l.matchEndIdx = l.matchIndex + matchLength;
            }
            brk = (p + matchLength == iend); // equal: no way to know if inf or sup
        } else if (matchLength > l.bestLength) {
            if (matchLength > l.matchEndIdx - l.matchIndex)
                l.matchEndIdx = l.matchIndex + matchLength;
            l.bestLength = matchLength;
            if (l.grp && l.j != 0 && l.mnum >= GRP_MC) { // (the group's price step works on the matches kept in registers; a leader's
                                                         // further matches go to the workspace and its prices the one-lane way)
                abort = true;
                return true;
            }
            grpMSet(l, l.mnum, (l.wk_current - l.matchIndex) + REP_MOVE, matchLength);
            l.last_m_off = (l.wk_current - l.matchIndex) + REP_MOVE;
            l.last_m_len = matchLength;
            l.mnum++;
            if ((matchLength > OPT_NUM) | (p + matchLength == iend))
                brk = true; // drop, to preserve bt consistency
        }
        if (!brk) {
            ended = false;
            const bool smaller = mByte < pByte; // match[matchLength] < p[matchLength]
            const U32 ptr = smaller ? l.smallerPtr : l.largerPtr;
            if (ptr != SM_NOPTR) {
                if (!UPD && l.rec)
                    grpRecPut(l, ptr, l.matchIndex);
                else
                    bt[ptr] = l.matchIndex;
            }
            const bool low = l.matchIndex <= l.btLow;
            if (smaller) {
                l.clSmaller = matchLength;
                l.smallerPtr = low ? SM_NOPTR : nextPtr + 1;
            } else {
                l.clLarger = matchLength;
                l.largerPtr = low ? SM_NOPTR : nextPtr;
            }
            if (low)
                ended = true;
            else
                l.matchIndex = smaller ? childSmaller : childLarger;
        }
    }
    if (ended) {
        if (l.smallerPtr != SM_NOPTR) {
            if (!UPD && l.rec)
                grpRecPut(l, l.smallerPtr, 0);
            else
                bt[l.smallerPtr] = 0;
        }
        if (l.largerPtr != SM_NOPTR) {
            if (!UPD && l.rec)
                grpRecPut(l, l.largerPtr, 0);
            else
                bt[l.largerPtr] = 0;
        }
    } else if (!UPD)
        grpWalkIssue(l, src, iend, bt, btMask);
    return ended;
}

// The COMMON level of a walk, without a loop and almost without branches: the node was read ahead and the position and the
// match differ inside the 8 bytes read.  Everything else -- no node left (the walk closes), 8 equal bytes, a node near the end of
// the block, a full record -- clears wk_pre and is left to grpWalkLevel, which segment E runs ONCE per trip behind a row of
// these (a level of grpWalkLevel is ~650 instructions, half of them control flow; the trip is issue-bound).
ZFN void grpWalkFastLevel(GLane &l, const BYTE *src, const BYTE *iend, U32 *bt, U32 btMask)
{
    const U64 d = l.wk_pb ^ l.wk_mb;
    if (d == 0 || (l.grp && (l.nrec + 3 > GRP_RC || (l.j != 0 && l.mnum >= GRP_MC)))) {
        l.wk_pre = false;
        return;
    }
    l.nbCompares--;
    const U32 nextPtr = 2 * (l.matchIndex & btMask);
    const U32 ml0 = l.clSmaller < l.clLarger ? l.clSmaller : l.clLarger;
    const U32 childLarger = (U32)l.wk_pair;          // nextPtr[0]
    const U32 childSmaller = (U32)(l.wk_pair >> 32); // nextPtr[1]
    const U32 sh = (U32)__builtin_ctzll(d) & ~7u;
    const U32 pByte = (U32)(l.wk_pb >> sh) & 0xFF;
    const U32 mByte = (U32)(l.wk_mb >> sh) & 0xFF;
    const U32 matchLength = ml0 + (sh >> 3);
    bool brk = false;
    if (matchLength > l.bestLength) {
        if (matchLength > l.matchEndIdx - l.matchIndex)
            l.matchEndIdx = l.matchIndex + matchLength;
        l.bestLength = matchLength;
        grpMSet(l, l.mnum, (l.wk_current - l.matchIndex) + REP_MOVE, matchLength);
        l.last_m_off = (l.wk_current - l.matchIndex) + REP_MOVE;
        l.last_m_len = matchLength;
        l.mnum++;
        brk = matchLength > OPT_NUM; // (the position and the match differ: the end of the block is not reached)
    }
    bool over = brk;
    if (!brk) {
        const bool smaller = mByte < pByte; // match[matchLength] < p[matchLength]
        const U32 ptr = smaller ? l.smallerPtr : l.largerPtr;
        if (ptr != SM_NOPTR) {
            if (l.rec)
                grpRecPut(l, ptr, l.matchIndex);
            else
                bt[ptr] = l.matchIndex;
        }
        const bool low = l.matchIndex <= l.btLow;
        l.clSmaller = smaller ? matchLength : l.clSmaller;
        l.clLarger = smaller ? l.clLarger : matchLength;
        const U32 np = low ? SM_NOPTR : (smaller ? nextPtr + 1 : nextPtr);
        l.smallerPtr = smaller ? np : l.smallerPtr;
        l.largerPtr = smaller ? l.largerPtr : np;
        l.matchIndex = low ? l.matchIndex : (smaller ? childSmaller : childLarger);
        over = low;
    }
    if (over) { // the walk closes: grpWalkLevel finds no node to visit and makes (records) the closing stores
        l.nbCompares = 0;
        l.wk_pre = false;
    } else
        grpWalkIssue(l, src, iend, bt, btMask);
}

// common length of p and q given their first 8 bytes (p + 8 <= iend): the loop of ZSTD_count only runs when all 8 are equal
ZFN U32 grpCount8(U64 pv, U64 qv, const BYTE *p, const BYTE *q, const BYTE *iend)
{
    const U64 d = pv ^ qv;
    if (d)
        return (U32)__builtin_ctzll(d) >> 3;
    return 8 + count(p + 8, q + 8, iend);
}

// The group lanes' form of grpRepsAndHash3 (below): the bytes every test starts from -- the sources of the (up to) three
// repcodes, the hash-3 candidate, the root node of the tree walk -- are asked for TOGETHER, then looked at: one round trip
// instead of up to five.  Same decisions in the same order.
ZFN bool grpRepsAndHash3Pre(GLane &l, const BYTE *src, const BYTE *iend, U32 minMatch, U32 mls, U32 sufficient_len, U32 btMask)
{
    const CParams &cp = l.w.cp;
    const BYTE *const p = src + (l.wk_current - l.w.idx0);
    const U64 pv = l.p8; // (segment B read it at this position)
    l.clSmaller = l.clLarger = 0;
    const U32 dictLimit = l.w.dictLimit;
    l.btLow = (btMask >= l.wk_current) ? 0 : l.wk_current - btMask;
    const U32 maxDistance = 1u << cp.windowLog;
    const U32 windowLow = (l.wk_current - dictLimit > maxDistance) ? l.wk_current - maxDistance : dictLimit;
    l.lowLimit = windowLow ? windowLow : 1; // matchLow
    l.smallerPtr = 2 * (l.wk_current & btMask);
    l.largerPtr = l.smallerPtr + 1;
    l.matchEndIdx = l.wk_current + 8 + 1;
    l.mnum = 0;
    l.nbCompares = 1u << cp.searchLog;
    l.bestLength = minMatch - 1; // lengthToBeat - 1
    // ---- everything is asked for ----
    U32 roff[3];
    bool rok[3];
    U64 rv[3] = {0, 0, 0};
#pragma unroll
    for (U32 k = 0; k < 3; ++k) {
        const U32 repCode = l.q_ll0 + k;
        roff[k] = (repCode == REP_NUM) ? (l.q_rep0 - 1) : grpSel3(l.q_rep0, l.q_rep1, l.q_rep2, repCode);
        rok[k] = (roff[k] - 1 /* intentional overflow, discards 0 and -1 */ < l.wk_current - dictLimit) && (l.wk_current - roff[k] >= windowLow);
        if (rok[k])
            rv[k] = read64(p - roff[k]);
    }
    const bool ok3 = (mls == 3) && (l.mi3 >= l.lowLimit) && (l.wk_current - l.mi3 < (1u << 18));
    U64 m3v = 0;
    if (ok3)
        m3v = read64(src + (l.mi3 - l.w.idx0));
    l.matchIndex = l.mi0;
    grpWalkIssue(l, src, iend, l.w.chainTable, btMask);
    // ---- and looked at ----
    bool done = false;
#pragma unroll
    for (U32 k = 0; k < 3; ++k) {
        if (done)
            break;
        U32 repLen = 0;
        if (rok[k]) {
            const U32 d = (U32)(pv ^ rv[k]);
            if ((minMatch == 3 ? (d << 8) : d) == 0)
                repLen = grpCount8(pv, rv[k], p, p - roff[k], iend);
        }
        if (repLen > l.bestLength) {
            l.bestLength = repLen;
            grpMSet(l, l.mnum, k, repLen); // (repCode - ll0)
            l.last_m_off = k;
            l.last_m_len = repLen;
            l.mnum++;
            if ((repLen > sufficient_len) | (p + repLen == iend))
                done = true; // best possible
        }
    }
    if (!done && (mls == 3) && (l.bestLength < mls) && ok3) { // HC3 match finder
        const U32 mlen = grpCount8(pv, m3v, p, src + (l.mi3 - l.w.idx0), iend);
        if (mlen >= mls) {
            l.bestLength = mlen;
            grpMSet(l, 0, (l.wk_current - l.mi3) + REP_MOVE, mlen);
            l.last_m_off = (l.wk_current - l.mi3) + REP_MOVE;
            l.last_m_len = mlen;
            l.mnum = 1;
            if ((mlen > sufficient_len) | (p + mlen == iend))
                done = true;
        }
    }
    return done;
}

// repcodes and the hash-3 probe of ZSTD_insertBtAndGetAllMatches for the lane's request (wk_current, q_*), and the set-up of
// the tree walk.  Returns true when the request is answered without a walk ("done" in the library: best possible match).
// mi3 < 0xFFFFFFFF: the hash-3 table's answer is already known (group lanes); else it is read (and the table brought up to date).
ZFN bool grpRepsAndHash3(GLane &l, const BYTE *src, const BYTE *iend, U32 minMatch, U32 mls, U32 sufficient_len, U32 btMask, bool group)
{
    const CParams &cp = l.w.cp;
    const BYTE *const p = src + (l.wk_current - l.w.idx0);
    const U64 pv = (l.p8_pos == l.wk_current - l.w.idx0) ? l.p8 : read64(p); // (p <= iend - 8)
    l.clSmaller = l.clLarger = 0;
    const U32 dictLimit = l.w.dictLimit;
    l.btLow = (btMask >= l.wk_current) ? 0 : l.wk_current - btMask;
    const U32 maxDistance = 1u << cp.windowLog;
    const U32 windowLow = (l.wk_current - dictLimit > maxDistance) ? l.wk_current - maxDistance : dictLimit;
    l.lowLimit = windowLow ? windowLow : 1; // matchLow
    l.smallerPtr = 2 * (l.wk_current & btMask);
    l.largerPtr = l.smallerPtr + 1;
    l.matchEndIdx = l.wk_current + 8 + 1;
    l.mnum = 0;
    l.nbCompares = 1u << cp.searchLog;
    l.bestLength = minMatch - 1; // lengthToBeat - 1
    bool done = false;
    {
        const U32 lastR = REP_NUM + l.q_ll0;
        for (U32 repCode = l.q_ll0; repCode < lastR; repCode++) {
            const U32 repOffset = (repCode == REP_NUM) ? (l.q_rep0 - 1) : grpSel3(l.q_rep0, l.q_rep1, l.q_rep2, repCode);
            const U32 repIndex = l.wk_current - repOffset;
            U32 repLen = 0;
            if (repOffset - 1 /* intentional overflow, discards 0 and -1 */ < l.wk_current - dictLimit) {
                if ((repIndex >= windowLow) & ((minMatch == 3 ? ((U32)pv << 8) : (U32)pv) == readMINMATCH(p - repOffset, minMatch)))
                    repLen = count(p + minMatch, p + minMatch - repOffset, iend) + minMatch;
            }
            if (repLen > l.bestLength) {
                l.bestLength = repLen;
                grpMSet(l, l.mnum, repCode - l.q_ll0, repLen);
                l.last_m_off = repCode - l.q_ll0;
                l.last_m_len = repLen;
                l.mnum++;
                if ((repLen > sufficient_len) | (p + repLen == iend)) {
                    done = true; // best possible
                    break;
                }
            }
        }
    }
    if (!done && (mls == 3) && (l.bestLength < mls)) { // HC3 match finder
        U32 matchIndex3;
        if (group)
            matchIndex3 = l.mi3;
        else { // ZSTD_insertAndFindFirstIndexHash3
            const U32 h3 = hash3((U32)pv, l.w.hashLog3);
            for (U32 idx = l.nextToUpdate3; idx < l.wk_current; ++idx)
                l.w.hashTable3[hash3(read32(src + (idx - l.w.idx0)), l.w.hashLog3)] = idx;
            l.nextToUpdate3 = l.wk_current;
            matchIndex3 = l.w.hashTable3[h3];
        }
        if ((matchIndex3 >= l.lowLimit) & (l.wk_current - matchIndex3 < (1u << 18))) {
            const BYTE *const match = src + (matchIndex3 - l.w.idx0);
            const U32 mlen = count(p, match, iend);
            if (mlen >= mls) {
                l.bestLength = mlen;
                grpMSet(l, 0, (l.wk_current - matchIndex3) + REP_MOVE, mlen);
                l.last_m_off = (l.wk_current - matchIndex3) + REP_MOVE;
                l.last_m_len = mlen;
                l.mnum = 1;
                if ((mlen > sufficient_len) | (p + mlen == iend)) {
                    if (!group)
                        l.w.nextToUpdate = l.wk_current + 1; // skip insertion
                    done = true;
                }
            }
        }
    }
    return done;
}

// ZSTD_compressBlock_opt_generic for a group of G lanes; lanes = the group's lane states (device: the calling lane's own),
// sh = the group's exchange record.  rep[] (the block's repcodes) is the leader's.  Returns the last literals (leader).
template <int G> ZFN U32 compressBlockOptGrp(GLane *lanes, GrpX &sh, U32 rep[3], const BYTE *src, U32 srcSize, int optLevel)
{
    static_assert(G >= 1 && G <= (int)GRP_MAX, "group size");
    const BYTE *const iend = src + srcSize;
    const U32 ilimit_off = srcSize - 8; // ilimit = iend - 8 (srcSize >= 8: smaller blocks never get here)
    const CParams cp = lanes[0].w.cp;
    const U32 sufficient_len = cp.targetLength < OPT_NUM - 1 ? cp.targetLength : OPT_NUM - 1;
    const U32 minMatch = (cp.minMatch == 3) ? 3 : 4;
    const U32 mls = cp.minMatch <= 3 ? 3 : (cp.minMatch == 4 ? 4 : (cp.minMatch == 5 ? 5 : 6));
    const U32 btMask = (1u << (cp.chainLog - 1)) - 1;
    const bool grpOk = (optLevel == 2) && grpEligible(cp, srcSize);
    const U32 gmax = (U32)G < minMatch ? (U32)G : minMatch;

    ZS_GRP_EACH(l)
    l.inChunk = false;
    l.wide = grpWide(cp, srcSize);
    l.rec = false;
    l.grp = false;
    l.nrec = 0;
    l.gstatus = GS_OK;
    l.p8 = 0;
    l.p8_pos = 0xFFFFFFFFu;
    l.wk_pre = false;
    l.adv = 1;
    l.g = 0;
    l.lastSequence.price = 0;
    l.lastSequence.off = l.lastSequence.mlen = l.lastSequence.litlen = 0;
    l.lastSequence.rep[0] = l.lastSequence.rep[1] = l.lastSequence.rep[2] = 0;
    if (l.j == 0) {
        l.ip = l.anchor = 0;
        l.cur = l.last_pos = 0;
        l.rep0 = rep[0];
        l.rep1 = rep[1];
        l.rep2 = rep[2];
        l.nextToUpdate3 = l.w.nextToUpdate;
        rescaleFreqs(l.w, src, srcSize, optLevel);
        grpPublishBases(sh, l.w);
        l.ip += (l.w.idx0 == l.w.dictLimit);
        l.state = ST_FIND_FIRST;
        sh.done = 0;
        sh.g = 0;
    } else
        l.state = ST_F_IDLE;
    ZS_GRP_END

#if defined(__HIP_DEVICE_COMPILE__) && defined(ZS_GRP_PROF)
    unsigned long long zs_prof[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long zs_prof_t = __builtin_readcyclecounter();
#endif
    for (;;) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(ZS_GRP_PROF)
        zs_prof[9]++;
#endif
#ifdef ZS_GRP_STATS
        g_grp_trips++;
#endif
        // ================= segment A: the leader's chunk-level steps; the plan of a group trip =================
        ZS_GRP_EACH(l)
        ZS_OPAQUE(l.state);
        OptWs &w = l.w;
        Optimal *const opt = w.opt;
        if (l.j == 0)
            sh.g = 0;
        if (l.state == ST_STORE) do { // a few sequences of the finished chunk per trip
            // the entries of this trip: four positions of the path from fw[], their heads asked for together, then stored in order
            U32 pos[GRP_STORE_SEQS];
            U32 llen_[GRP_STORE_SEQS], mlen_[GRP_STORE_SEQS], off_[GRP_STORE_SEQS];
            // (storePos: index into fw[], the path's positions in forward order up to storeEnd; lastSequence follows them)
            const U64 four = read64((const BYTE *)(w.fw + l.storePos)); // (four links in one read; the array has slack behind its end)
            U32 p = l.storePos;
#pragma unroll
            for (U32 k = 0; k < GRP_STORE_SEQS; ++k) {
                const U32 idx = l.storePos + k;
                pos[k] = idx < l.storeEnd ? (U32)(four >> (16 * k)) & 0xFFFFu : (idx == l.storeEnd ? GRP_ST_LAST : GRP_ST_DONE);
            }
            p = l.storePos + GRP_STORE_SEQS;
            const bool chunkStored = p > l.storeEnd; // lastSequence was among the four
#pragma unroll
            for (U32 k = 0; k < GRP_STORE_SEQS; ++k) {
                llen_[k] = l.lastSequence.litlen;
                mlen_[k] = l.lastSequence.mlen;
                off_[k] = l.lastSequence.off;
                if (pos[k] < GRP_ST_LAST) {
                    llen_[k] = opt[pos[k]].litlen;
                    mlen_[k] = opt[pos[k]].mlen;
                    off_[k] = opt[pos[k]].off;
                }
            }
            // where the sequences' literal runs start; the common case -- runs of at most GRP_ST_LITS literals, away from the end of
            // the block -- is stored without a dependent chain: the runs' bytes in one round trip, the literals' counts in another
            U32 anc[GRP_STORE_SEQS];
            bool isSeq[GRP_STORE_SEQS];
            bool fast = true;
            {
                U32 a = l.anchor;
#pragma unroll
                for (U32 k = 0; k < GRP_STORE_SEQS; ++k) {
                    isSeq[k] = pos[k] != GRP_ST_DONE && mlen_[k] != 0;
                    anc[k] = a;
                    if (isSeq[k]) {
                        fast = fast && llen_[k] <= GRP_ST_LITS && a <= ilimit_off; // (a + 8 <= srcSize: the run is read as 8 bytes)
                        a += llen_[k] + mlen_[k];
                    }
                }
            }
            if (fast) {
                U64 lb[GRP_STORE_SEQS];
#pragma unroll
                for (U32 k = 0; k < GRP_STORE_SEQS; ++k)
                    lb[k] = (isSeq[k] && llen_[k]) ? read64(src + anc[k]) : 0;
                U32 byt[GRP_STORE_SEQS * GRP_ST_LITS], frq[GRP_STORE_SEQS * GRP_ST_LITS];
                bool on[GRP_STORE_SEQS * GRP_ST_LITS];
#pragma unroll
                for (U32 k = 0; k < GRP_STORE_SEQS; ++k)
#pragma unroll
                    for (U32 u = 0; u < GRP_ST_LITS; ++u) {
                        const U32 i = k * GRP_ST_LITS + u;
                        on[i] = isSeq[k] && u < llen_[k];
                        byt[i] = (U32)(lb[k] >> (8 * u)) & 0xFF;
                        frq[i] = on[i] ? w.litFreq[byt[i]] : 0;
                    }
                // ZSTD_updateStats: litFreq[literal] += ZSTD_LITFREQ_ADD, literal after literal (equal bytes of the batch see each other:
                // the later store carries the sum)
#pragma unroll
                for (U32 i = 0; i < GRP_STORE_SEQS * GRP_ST_LITS; ++i)
                    if (on[i]) {
                        U32 c = 1;
#pragma unroll
                        for (U32 jj = 0; jj < i; ++jj)
                            c += (on[jj] && byt[jj] == byt[i]) ? 1u : 0u;
                        w.litFreq[byt[i]] = frq[i] + LITFREQ_ADD * c;
                    }
#pragma unroll
                for (U32 k = 0; k < GRP_STORE_SEQS; ++k) {
                    if (isSeq[k]) {
                        const U32 llen = llen_[k], mlen = mlen_[k], offCode = off_[k];
                        w.litSum += llen * LITFREQ_ADD;
                        w.litLengthFreq[LLcode(llen)]++;
                        w.litLengthSum++;
                        w.offCodeFreq[highbit32(offCode + 1)]++;
                        w.offCodeSum++;
                        w.matchLengthFreq[MLcode(mlen - MINMATCH)]++;
                        w.matchLengthSum++;
                        // ZSTD_storeSeq: the run's bytes in one (wider) store -- what lies behind the run is overwritten by the next
                        if (llen)
                            memcpy(w.lits + w.nLits, &lb[k], 8);
                        w.nLits += llen;
                        Seq &sq = w.seqs[w.nSeq++];
                        sq.offCode = offCode;
                        sq.litLength = llen;
                        sq.matchLength = mlen;
                        l.anchor += llen + mlen;
                        l.ip = l.anchor;
                    } else if (pos[k] != GRP_ST_DONE) // only literals => must be last "sequence", actually starting a new stream of sequences
                        l.ip = l.anchor + llen_[k];
                }
            } else {
#pragma unroll 1
                for (U32 k = 0; k < GRP_STORE_SEQS; ++k) {
                    if (pos[k] == GRP_ST_DONE)
                        break;
                    if (mlen_[k] == 0) { // only literals => must be last "sequence", actually starting a new stream of sequences
                        l.ip = l.anchor + llen_[k];
                    } else {
                        updateStats(w, llen_[k], src + l.anchor, off_[k], mlen_[k]);
                        storeSeq(w, llen_[k], src + l.anchor, off_[k], mlen_[k]);
                        l.anchor += llen_[k] + mlen_[k];
                        l.ip = l.anchor;
                    }
                }
            }
            l.storePos = p;
            if (chunkStored) {
                setBasePrices(w, optLevel);
                grpPublishBases(sh, w);
                l.state = ST_FIND_FIRST;
            }
        } while (0);
        ZS_GRP_TICK(10);
        if (l.state == ST_CUR_NEXT) do {
            l.cur += l.adv;
            l.adv = 1;
            if (l.cur > l.last_pos) { // the forward loop ran out
                l.lastSequence = opt[l.last_pos];
                const U32 tl = l.lastSequence.litlen + l.lastSequence.mlen;
                l.cur = l.last_pos > tl ? l.last_pos - tl : 0; // single sequence, and it starts before `ip`
                l.state = ST_CHUNK_END;
                break;
            }
            // the trip's plan: lane i takes position cur + i as long as that position cannot be the chunk's last one and a match
            // may still start there; the leader's own position goes the group way when nothing has to be inserted before it
            const U32 inr0 = l.ip + l.cur;
            const U32 q0 = inr0 + w.idx0;
            U32 g = 1;
            bool plain0 = grpOk && inr0 <= ilimit_off && l.cur < l.last_pos && w.nextToUpdate == q0 && l.nextToUpdate3 <= q0 &&
                          q0 - l.nextToUpdate3 <= 1;
            if (plain0) {
                while (g < gmax && l.cur + g < l.last_pos && inr0 + g <= ilimit_off)
                    ++g;
                // the hash-3 table is brought up to the leader's position (the library does it at the next probe; a probe only ever
                // looks for positions in front of its own, so doing it early changes no answer)
                const U32 n = q0 - l.nextToUpdate3;
                sh.pend_n = n;
                for (U32 t = 0; t < n; ++t) {
                    const U32 idx = l.nextToUpdate3 + t;
                    const U32 hh = hash3(read32(src + (idx - w.idx0)), w.hashLog3);
                    w.hashTable3[hh] = idx;
                    sh.pend_h3 = hh;
                }
                l.nextToUpdate3 = q0;
            }
            sh.g = g | (plain0 ? 256u : 0u);
            sh.cur = l.cur;
            sh.last_pos = l.last_pos;
            sh.ip = l.ip;
            l.g = g;
            l.state = ST_G_BEGIN;
        } while (0);
        ZS_GRP_TICK(11);
        if (l.state == ST_FIND_FIRST) do {
            if (!(srcSize >= 8 && l.ip < ilimit_off)) {
                l.state = ST_DONE;
                break;
            }
            l.p8 = read64(src + l.ip); // (ip < srcSize - 8)
            l.p8_pos = l.ip;
            l.q_litlen = l.ip - l.anchor;
            l.q_ll0 = !l.q_litlen;
            l.q_current = l.ip + w.idx0;
            l.q_rep0 = l.rep0;
            l.q_rep1 = l.rep1;
            l.q_rep2 = l.rep2;
            l.inChunk = false;
            // the request of a chunk's first position goes the recorded-walk way as well (the leader alone), unless skipped
            // positions have to be inserted first
            const U32 q0 = l.q_current;
            if (grpOk && w.nextToUpdate == q0 && l.nextToUpdate3 <= q0 && q0 - l.nextToUpdate3 <= 1) {
                const U32 n = q0 - l.nextToUpdate3;
                sh.pend_n = n;
                for (U32 t = 0; t < n; ++t) {
                    const U32 idx = l.nextToUpdate3 + t;
                    const U32 hh = hash3(read32(src + (idx - w.idx0)), w.hashLog3);
                    w.hashTable3[hh] = idx;
                    sh.pend_h3 = hh;
                }
                l.nextToUpdate3 = q0;
                l.g = 1;
                l.state = ST_G_FIRST;
            } else
                l.state = ST_GETM_BEGIN;
        } while (0);
        ZS_GRP_END

        ZS_GRP_TICK(0);
        // ================= segment B: the lanes of the trip read what their positions need =================
        ZS_GRP_EACH(l)
        OptWs &w = l.w;
        Optimal *const opt = w.opt;
        if (l.state == ST_F_IDLE && (sh.g & 255u) > l.j) { // a follower joins
            l.g = sh.g & 255u;
            l.cur = sh.cur;
            l.last_pos = sh.last_pos;
            l.ip = sh.ip;
            w.priceType = sh.priceType;
            w.litSumBasePrice = sh.litSumBP;
            w.litLengthSumBasePrice = sh.llSumBP;
            w.matchLengthSumBasePrice = sh.mlSumBP;
            w.offCodeSumBasePrice = sh.ocSumBP;
            l.state = ST_G_BEGIN;
        }
        if (l.state == ST_G_FIRST) { // (leader) a chunk's first position: no price-table entry to finish, only the tables' answers
            l.h = mls == 5 ? hash5(l.p8, cp.hashLog) : mls == 6 ? hash6(l.p8, cp.hashLog) : hash4((U32)l.p8, cp.hashLog);
            l.h3 = hash3((U32)l.p8, w.hashLog3);
            l.mi0 = w.hashTable[l.h];
            l.mi3 = w.hashTable3[l.h3];
            sh.h[0] = l.h;
            sh.h3[0] = l.h3;
            sh.wdone[0] = 0;
        }
        if (l.state == ST_G_BEGIN) {
            l.g_cur = l.cur + l.j;
            const U32 inr = l.ip + l.g_cur;
            l.oc = opt[l.g_cur];
            if (l.j == 0)
                l.op = opt[l.g_cur - 1];
            const U32 lit_byte = src[inr - 1];
            const bool at = inr <= ilimit_off; // (a follower's position always is)
            if (at) {
                l.p8 = read64(src + inr);
                l.p8_pos = inr;
            }
            l.lit_freq = (w.priceType == zop_predef) ? 0 : w.litFreq[lit_byte];
            l.pr0 = l.pr1 = l.pr2 = 0;
            // (the entry a match came from lies at least minMatch positions back: final since an earlier trip; an entry that only
            // holds MAX_PRICE has a stale mlen -- the literal step below replaces it, its repcodes are never looked at)
            if (l.oc.mlen != 0 && l.oc.mlen <= l.g_cur) {
                const U32 prev = l.g_cur - l.oc.mlen;
                l.pr0 = opt[prev].rep[0];
                l.pr1 = opt[prev].rep[1];
                l.pr2 = opt[prev].rep[2];
            }
            l.h = l.h3 = l.mi0 = l.mi3 = 0;
            if (at && (sh.g >> 8)) {
                l.h = mls == 5 ? hash5(l.p8, cp.hashLog) : mls == 6 ? hash6(l.p8, cp.hashLog) : hash4((U32)l.p8, cp.hashLog);
                l.h3 = hash3((U32)l.p8, w.hashLog3);
                l.mi0 = w.hashTable[l.h];
                l.mi3 = w.hashTable3[l.h3];
                sh.h[l.j] = l.h;
                sh.h3[l.j] = l.h3;
            }
            sh.wdone[l.j] = 0;
            // the prices the trip's first targets hold now (nothing writes them before segment I)
            l.oldp_t0 = l.cur + minMatch < l.last_pos + 1 ? l.cur + minMatch : l.last_pos + 1;
#pragma unroll
            for (U32 s = 0; s < GRP_PT; ++s) {
                const U32 t = l.oldp_t0 + s;
                l.oldp[s] = (t <= l.last_pos) ? opt[t].price : MAX_PRICE;
            }
        }
        ZS_GRP_END

        ZS_GRP_TICK(1);
        // ================= segments C0 .. C(G-1): the literal step, chained through the trip's positions =================
#pragma unroll
        for (int s = 0; s < G; ++s) {
            ZS_GRP_EACH(l)
            OptWs &w = l.w;
            if (l.state == ST_G_BEGIN && (int)l.j == s) {
                if (s > 0) {
                    l.op.price = sh.oc_price;
                    l.op.mlen = sh.oc_mlen;
                    l.op.litlen = sh.oc_litlen;
                    l.op.rep[0] = sh.oc_rep[0];
                    l.op.rep[1] = sh.oc_rep[1];
                    l.op.rep[2] = sh.oc_rep[2];
                }
                {
                    const U32 litlen = (l.op.mlen == 0) ? l.op.litlen + 1 : 1;
                    // rawLiteralsCost(src + inr - 1, 1)
                    const U32 lit_cost = (w.priceType == zop_predef) ? 6 * BITCOST_MULTIPLIER : w.litSumBasePrice - weight(l.lit_freq, optLevel);
                    const int price = l.op.price + (int)lit_cost + (int)litLengthPrice(litlen, w, optLevel) - (int)litLengthPrice(litlen - 1, w, optLevel);
                    if (price <= l.oc.price) {
                        l.oc.mlen = 0;
                        l.oc.off = 0;
                        l.oc.litlen = litlen;
                        l.oc.price = price;
                    }
                }
                if (l.oc.mlen != 0) {
                    U32 pr[3] = {l.pr0, l.pr1, l.pr2};
                    updateRep(l.oc.rep, pr, l.oc.off, l.oc.litlen == 0);
                } else {
                    l.oc.rep[0] = l.op.rep[0];
                    l.oc.rep[1] = l.op.rep[1];
                    l.oc.rep[2] = l.op.rep[2];
                }
                w.opt[l.g_cur] = l.oc;
                w.bk[l.g_cur] = l.oc.litlen + l.oc.mlen; // (what the chunk end's walk back reads: a compact copy)
                if (s + 1 < G) {
                    sh.oc_price = l.oc.price;
                    sh.oc_mlen = l.oc.mlen;
                    sh.oc_litlen = l.oc.litlen;
                    sh.oc_rep[0] = l.oc.rep[0];
                    sh.oc_rep[1] = l.oc.rep[1];
                    sh.oc_rep[2] = l.oc.rep[2];
                }
            }
            ZS_GRP_END
        }

        ZS_GRP_TICK(2);
        // ================= segment D: match requests: repcodes, hash-3 probe, walk set-up =================
        ZS_GRP_EACH(l)
        OptWs &w = l.w;
        Optimal *const opt = w.opt;
        if (l.state == ST_G_BEGIN) do {
            const U32 inr = l.ip + l.g_cur;
            if (l.j == 0) {
                if (inr > ilimit_off) { // last match must start at a minimum distance of 8 from oend
                    l.state = ST_CUR_NEXT;
                    break;
                }
                if (l.cur == l.last_pos) { // `break` of the forward loop
                    l.lastSequence = l.oc;
                    const U32 tl = l.lastSequence.litlen + l.lastSequence.mlen;
                    l.cur = l.last_pos > tl ? l.last_pos - tl : 0;
                    l.state = ST_CHUNK_END;
                    break;
                }
                if ((optLevel == 0) && (opt[l.cur + 1].price <= l.oc.price + (int)(BITCOST_MULTIPLIER / 2))) {
                    l.state = ST_CUR_NEXT; // skip unpromising positions
                    break;
                }
            }
            l.q_ll0 = (l.oc.mlen != 0);
            l.q_litlen = (l.oc.mlen == 0) ? l.oc.litlen : 0;
            l.cur_litlen_back = (l.oc.mlen == 0) ? l.oc.litlen : 0; // what `cur -= ...` of the early chunk end needs
            l.basePrice = (U32)l.oc.price + litLengthPrice(0, w, optLevel);
            l.q_current = inr + w.idx0;
            l.q_rep0 = l.oc.rep[0];
            l.q_rep1 = l.oc.rep[1];
            l.q_rep2 = l.oc.rep[2];
            l.inChunk = true;
            if (!(sh.g >> 8)) { // (leader alone) the one-lane way
                l.state = ST_GETM_BEGIN;
                break;
            }
            l.state = ST_G_FIRST; // (= the request below)
        } while (0);
        if (l.state == ST_G_FIRST) do {
            l.wk_current = l.q_current;
            l.gstatus = GS_OK;
            // the hash-3 table's answer for this position: what the table held, then the positions the leader has just inserted,
            // then the trip's positions in front of this one (the latest position with the same hash wins)
            if (sh.pend_n && sh.pend_h3 == l.h3)
                l.mi3 = l.q_current - l.j - 1;
            for (U32 i = 0; i < l.j; ++i) {
                if (sh.h3[i] == l.h3)
                    l.mi3 = l.q_current - l.j + i;
                if (sh.h[i] == l.h)
                    l.gstatus = GS_ANOMALY; // same tree as a position in front: the walk would miss that position
            }
            const bool done = grpRepsAndHash3Pre(l, src, iend, minMatch, mls, sufficient_len, btMask);
            if (done)
                l.gstatus = GS_ANOMALY; // answered without a walk and without an insertion: the one-lane path's business
            l.rec = true;
            l.grp = true;
            l.nrec = 0;
            if (l.gstatus != GS_OK) {
                sh.wdone[l.j] = 1 + GS_ANOMALY;
                l.state = ST_G_WAIT;
            } else
                l.state = ST_WALK;
        } while (0);
        // ---- the one-lane path (zs_opt_sm.h, state for state) ----
        if (l.state == ST_GETM_BEGIN) do {
            l.rec = false;
            l.grp = false;
            if (l.q_current < w.nextToUpdate) { // skipped area
                l.nbMatches = 0;
                l.state = ST_AFTER_MATCHES;
                break;
            }
            l.upd_idx = w.nextToUpdate;
            l.state = ST_UPD_BEGIN;
        } while (0);
        for (U32 it_ = 0; it_ < 2 && (l.state == ST_UPD_BEGIN || l.state == ST_UPD_WALK); ++it_) { // skipped positions: ZSTD_updateTree
            if (l.state == ST_UPD_BEGIN) do {
                if (!(l.upd_idx < l.q_current)) {
                    w.nextToUpdate = l.q_current;
                    l.state = ST_GETM_REP;
                    break;
                }
                // ZSTD_insertBt1 set-up for position upd_idx
                l.wk_current = l.upd_idx;
                const BYTE *const p = src + (l.wk_current - w.idx0);
                const U32 h = hashPtr(p, cp.hashLog, mls);
                l.matchIndex = w.hashTable[h];
                l.clSmaller = l.clLarger = 0;
                l.btLow = btMask >= l.wk_current ? 0 : l.wk_current - btMask;
                l.smallerPtr = 2 * (l.wk_current & btMask);
                l.largerPtr = l.smallerPtr + 1;
                l.lowLimit = w.dictLimit;
                l.matchEndIdx = l.wk_current + 8 + 1;
                l.bestLength = 8;
                l.nbCompares = 1u << cp.searchLog;
                w.hashTable[h] = l.wk_current;
                l.wk_pre = false;
                l.state = ST_UPD_WALK;
            } while (0);
            if (l.state == ST_UPD_WALK) {
                bool abort;
                if (grpWalkLevel<true>(l, src, iend, w.chainTable, btMask, abort)) {
                    U32 positions = 0;
                    if (l.bestLength > 384)
                        positions = l.bestLength - 384 < 192 ? l.bestLength - 384 : 192;
                    const U32 adv = l.matchEndIdx - (l.wk_current + 8);
                    l.upd_idx += positions > adv ? positions : adv;
                    l.state = ST_UPD_BEGIN;
                }
            }
        }
        if (l.state == ST_GETM_REP) do {
            l.wk_current = l.q_current;
            const BYTE *const p = src + (l.wk_current - w.idx0);
            const U64 pv = (l.p8_pos == l.wk_current - w.idx0) ? l.p8 : read64(p);
            const U32 h = mls == 5 ? hash5(pv, cp.hashLog) : mls == 6 ? hash6(pv, cp.hashLog) : hash4((U32)pv, cp.hashLog);
            l.matchIndex = w.hashTable[h];
            const bool done = grpRepsAndHash3(l, src, iend, minMatch, mls, sufficient_len, btMask, false);
            if (done) {
                l.nbMatches = l.mnum;
                l.state = ST_AFTER_MATCHES;
                break;
            }
            w.hashTable[h] = l.wk_current;
            grpWalkIssue(l, src, iend, w.chainTable, btMask);
            l.state = ST_WALK;
        } while (0);
        ZS_GRP_END

        ZS_GRP_TICK(3);
        // ================= segment E: tree walks (recorded for the lanes of a group trip) =================
        ZS_GRP_EACH(l)
        OptWs &w = l.w;
#pragma unroll
        for (U32 lv_ = 0; lv_ < GRP_WALK_LEVELS; ++lv_) // the common levels (read ahead, decided inside 8 bytes)
            if (l.state == ST_WALK && l.wk_pre)
                grpWalkFastLevel(l, src, iend, w.chainTable, btMask);
        if (l.state == ST_WALK && !l.wk_pre) { // everything else, once per trip: the end of a walk, long common prefixes ...
            bool abort;
            if (grpWalkLevel<false>(l, src, iend, w.chainTable, btMask, abort)) {
                if (l.grp) {
                    sh.nbm[l.j] = l.mnum;
                    sh.maxML[l.j] = l.last_m_len;
                    sh.maxOff[l.j] = l.last_m_off;
                    sh.mEnd[l.j] = l.matchEndIdx;
                    sh.qlit[l.j] = l.q_litlen;
                    sh.litback[l.j] = l.cur_litlen_back;
                    sh.wdone[l.j] = abort ? 1 + GS_ANOMALY : 1 + GS_OK;
                    l.nbMatches = l.mnum;
                    l.state = ST_G_WAIT;
                } else {
                    w.nextToUpdate = l.matchEndIdx - 8; // skip repetitive patterns
                    l.nbMatches = l.mnum;
                    l.state = ST_AFTER_MATCHES;
                }
            }
        }
        ZS_GRP_END

        ZS_GRP_TICK(4);
        // ================= segment F: the group is validated in order; valid lanes commit =================
        ZS_GRP_EACH(l)
        OptWs &w = l.w;
        if (l.state == ST_G_WAIT) do {
            bool all = true;
            for (U32 i = 0; i < l.g; ++i)
                all = all && (sh.wdone[i] != 0);
            if (!all)
                break;
            // (every lane of the group computes the same verdict from the same record)
            const U32 q0 = l.q_current - l.j;
            U32 v = 0, ntu = q0;
            int endLane = -1;
            bool bigLeader = false;
            for (U32 i = 0; i < l.g; ++i) {
                if (sh.wdone[i] != 1 + GS_OK)
                    break; // anomaly: this position and the ones behind it are redone
                if (i > 0 && q0 + i < ntu)
                    break; // skipped area (a long match in front moved nextToUpdate past it)
                v = i + 1;
                ntu = sh.mEnd[i] - 8;
                if (sh.nbm[i] > GRP_MC) { // (only a leader gets here with more matches than the registers hold)
                    bigLeader = true;
                    break;
                }
                if (l.inChunk && sh.nbm[i] && ((sh.maxML[i] > sufficient_len) || (l.cur + i + sh.maxML[i] >= OPT_NUM))) {
                    endLane = (int)i; // large match -> immediate encoding: the chunk ends at this position
                    break;
                }
            }
            U32 t1 = 0;
            bool anyMatch = false;
            for (U32 i = 0; i < v; ++i)
                if (sh.nbm[i]) {
                    anyMatch = true;
                    const U32 e = l.cur + i + sh.maxML[i];
                    t1 = e > t1 ? e : t1;
                }
            if (l.j < v) { // commit: the stores of ZSTD_insertBtAndGetAllMatches, the hash tables
                U32 *const bt = w.chainTable;
                for (U32 r = 0; r < l.nrec; ++r) {
                    U32 slot, val;
                    grpRecGet(l, r, slot, val);
                    bt[slot] = val;
                }
                w.hashTable[l.h] = l.q_current;
                bool later = false;
                for (U32 i = l.j + 1; i < v; ++i)
                    later = later || (sh.h3[i] == l.h3);
                if (!later)
                    w.hashTable3[l.h3] = l.q_current;
            }
            l.rec = false;
            l.grp = false;
            if (l.j == 0) {
                if (v == 0) { // the leader's own position is an anomaly: the one-lane way (nothing has been committed)
                    l.state = ST_GETM_BEGIN;
                    break;
                }
                w.nextToUpdate = ntu;
                l.nextToUpdate3 = q0 + v;
                if (!l.inChunk || bigLeader) { // a chunk's first position, or more matches than the group's price step takes: the
                    l.state = ST_AFTER_MATCHES; // parser's own checks and price loops follow (ST_AFTER_MATCHES below)
                    break;
                }
                if (endLane >= 0) {
                    l.lastSequence.mlen = sh.maxML[endLane];
                    l.lastSequence.off = sh.maxOff[endLane];
                    l.lastSequence.litlen = sh.qlit[endLane];
                    l.cur += (U32)endLane;
                    l.cur -= sh.litback[endLane]; // last sequence is actually only literals (may underflow)
                    l.last_pos = l.cur + l.lastSequence.litlen + l.lastSequence.mlen;
                    if (l.cur > OPT_NUM)
                        l.cur = 0; // underflow => first match
                    l.state = ST_CHUNK_END;
                    break;
                }
                l.adv = v;
                if (!anyMatch) {
                    l.state = ST_CUR_NEXT;
                    break;
                }
            } else if (l.j >= v || endLane >= 0 || !anyMatch || bigLeader) {
                l.state = ST_F_IDLE;
                break;
            }
#ifdef ZS_GRP_STATS
            if (l.j == 0) {
                g_grp_valid[v]++;
                g_grp_plan[l.g]++;
            }
#endif
            l.g_v = v;
            l.g_lp0 = l.last_pos;
            // targets from the first position a match can land on -- or from the first one behind the old end, if that comes
            // first: the sequential loop fills every position it passes over with MAX_PRICE
            l.tcur = l.cur + minMatch < l.last_pos + 1 ? l.cur + minMatch : l.last_pos + 1;
            l.t1 = t1;
            l.state = ST_G_PRICE;
        } while (0);
        // ---- one-lane path ----
        if (l.state == ST_AFTER_MATCHES) do {
            Optimal *const opt = w.opt;
            if (!l.inChunk) {
                if (!l.nbMatches) {
                    l.ip++;
                    l.state = ST_FIND_FIRST;
                    break;
                }
                opt[0].rep[0] = l.rep0;
                opt[0].rep[1] = l.rep1;
                opt[0].rep[2] = l.rep2;
                opt[0].mlen = 0;
                opt[0].litlen = l.q_litlen;
                opt[0].price = (int)litLengthPrice(l.q_litlen, w, optLevel);
                const U32 maxML = l.last_m_len;
                const U32 maxOffset = l.last_m_off;
                if (maxML > sufficient_len) { // large match -> immediate encoding
                    l.lastSequence.litlen = l.q_litlen;
                    l.lastSequence.mlen = maxML;
                    l.lastSequence.off = maxOffset;
                    l.cur = 0;
                    l.last_pos = l.lastSequence.litlen + l.lastSequence.mlen;
                    l.state = ST_CHUNK_END;
                    break;
                }
                l.pr_literalsPrice = (U32)opt[0].price + litLengthPrice(0, w, optLevel);
                for (U32 pos = 1; pos < minMatch; pos++)
                    opt[pos].price = MAX_PRICE;
                l.pr_pos = minMatch;
                l.pr_matchNb = 0;
                grpMGet(l, 0, l.pm_off, l.pm_len);
                l.state = ST_PRICE_FIRST;
                break;
            }
            if (!l.nbMatches) {
                l.state = ST_CUR_NEXT;
                break;
            }
            {
                const U32 maxML = l.last_m_len;
                if ((maxML > sufficient_len) || (l.cur + maxML >= OPT_NUM)) {
                    l.lastSequence.mlen = maxML;
                    l.lastSequence.off = l.last_m_off;
                    l.lastSequence.litlen = l.q_litlen;
                    l.cur -= l.cur_litlen_back; // last sequence is actually only literals (may underflow)
                    l.last_pos = l.cur + l.lastSequence.litlen + l.lastSequence.mlen;
                    if (l.cur > OPT_NUM)
                        l.cur = 0; // underflow => first match
                    l.state = ST_CHUNK_END;
                    break;
                }
            }
            l.pr_matchNb = 0;
            grpMGet(l, 0, l.pm_off, l.pm_len);
            l.pm_start = minMatch;
            l.pr_pos = l.pm_len; // mlen cursor of the downward scan
            l.state = ST_PRICE_CUR;
        } while (0);
        if (l.state == ST_PRICE_FIRST) do {
            Optimal *const opt = w.opt;
            U32 budget = SM_PRICE_STEPS;
            while (budget && l.pr_matchNb < l.nbMatches) {
                if (l.pr_pos <= l.pm_len) {
                    const U32 sequencePrice = l.pr_literalsPrice + getMatchPrice(l.pm_off, l.pr_pos, w, optLevel);
                    Optimal o; // (rep is set when the forward pass reaches the position)
                    o.price = (int)sequencePrice;
                    o.off = l.pm_off;
                    o.mlen = l.pr_pos;
                    o.litlen = l.q_litlen;
                    storeHead(opt[l.pr_pos], o);
                    l.pr_pos++;
                    budget--;
                } else {
                    l.pr_matchNb++;
                    if (l.pr_matchNb < l.nbMatches)
                        grpMGet(l, l.pr_matchNb, l.pm_off, l.pm_len);
                }
            }
            if (l.pr_matchNb >= l.nbMatches) {
                l.last_pos = l.pr_pos - 1;
                l.cur = 0; // the forward pass starts at 1 (ST_CUR_NEXT adds adv)
                l.adv = 1;
                l.state = ST_CUR_NEXT;
            }
        } while (0);
        if (l.state == ST_PRICE_CUR) do {
            Optimal *const opt = w.opt;
            U32 budget = SM_PRICE_STEPS;
            while (budget && l.pr_matchNb < l.nbMatches) {
                bool next = false;
                if (l.pr_pos >= l.pm_start) {
                    const U32 mlen = l.pr_pos;
                    const U32 pos = l.cur + mlen;
                    const int price = (int)(l.basePrice + getMatchPrice(l.pm_off, mlen, w, optLevel));
                    if ((pos > l.last_pos) || (price < opt[pos].price)) {
                        while (l.last_pos < pos) {
                            opt[l.last_pos + 1].price = MAX_PRICE;
                            l.last_pos++;
                        }
                        Optimal o;
                        o.price = price;
                        o.off = l.pm_off;
                        o.mlen = mlen;
                        o.litlen = l.q_litlen;
                        storeHead(opt[pos], o);
                    } else if (optLevel == 0)
                        next = true; // early update abort
                    l.pr_pos--;
                    budget--;
                    if (l.pr_pos < l.pm_start)
                        next = true;
                } else
                    next = true;
                if (next) {
                    l.pr_matchNb++;
                    if (l.pr_matchNb < l.nbMatches) {
                        l.pm_start = l.pm_len + 1; // matches[matchNb - 1].len + 1
                        grpMGet(l, l.pr_matchNb, l.pm_off, l.pm_len);
                        l.pr_pos = l.pm_len;
                    }
                }
            }
            if (l.pr_matchNb >= l.nbMatches) {
                l.adv = 1;
                l.state = ST_CUR_NEXT;
            }
        } while (0);
        ZS_GRP_END

        ZS_GRP_TICK(5);
        // ================= segment H: the lanes' prices for the next targets =================
        ZS_GRP_EACH(l)
        OptWs &w = l.w;
        if (l.state == ST_G_PRICE) {
            const bool has = l.nbMatches != 0;
#pragma unroll
            for (U32 s = 0; s < GRP_PT; ++s) {
                const U32 t = l.tcur + s;
                const U32 mlen = t - l.g_cur;
                int c = MAX_PRICE;
                if (has && t <= l.t1 && t >= l.g_cur + minMatch && mlen <= l.last_m_len) {
                    U32 off = l.m_off[0]; // the first match at least mlen long (lengths ascend)
#pragma unroll
                    for (U32 m = 1; m < GRP_MC; ++m)
                        if (m < l.nbMatches && l.m_len[m - 1] < mlen)
                            off = l.m_off[m];
                    c = (int)(l.basePrice + getMatchPrice(off, mlen, w, optLevel));
                    l.coff[s] = off;
                }
                l.cand[s] = c;
                sh.cand[l.j][s] = c;
            }
        }
        ZS_GRP_END

        ZS_GRP_TICK(6);
        // ================= segment I: per target the lanes' prices compete in lane order (strict <, like the sequential loop) =================
        ZS_GRP_EACH(l)
        OptWs &w = l.w;
        Optimal *const opt = w.opt;
        if (l.state == ST_G_PRICE) {
            int oldp[GRP_PT];
#pragma unroll
            for (U32 s = 0; s < GRP_PT; ++s) {
                const U32 t = l.tcur + s;
                oldp[s] = (l.tcur == l.oldp_t0) ? l.oldp[s] : ((t <= l.t1 && t <= l.g_lp0) ? opt[t].price : MAX_PRICE);
            }
#pragma unroll
            for (U32 s = 0; s < GRP_PT; ++s) {
                const U32 t = l.tcur + s;
                if (t <= l.t1) {
                    int best = oldp[s];
                    int winner = -1;
                    for (U32 i = 0; i < l.g_v; ++i) {
                        const int c = sh.cand[i][s];
                        if (c < best) {
                            best = c;
                            winner = (int)i;
                        }
                    }
                    if (winner == (int)l.j) {
                        Optimal o;
                        o.price = l.cand[s];
                        o.off = l.coff[s];
                        o.mlen = t - l.g_cur;
                        o.litlen = l.q_litlen;
                        storeHead(opt[t], o);
                    } else if (winner < 0 && l.j == 0 && t > l.g_lp0)
                        opt[t].price = MAX_PRICE; // (a position between the old end and a lane's first target)
                }
            }
            l.tcur += GRP_PT;
            if (l.tcur > l.t1) {
                if (l.j == 0) {
                    l.last_pos = l.t1 > l.g_lp0 ? l.t1 : l.g_lp0;
                    l.state = ST_CUR_NEXT;
                } else
                    l.state = ST_F_IDLE;
            }
        }
        ZS_GRP_TICK(12);
        if (l.state == ST_CHUNK_END) do {
            if (l.lastSequence.mlen != 0) {
                U32 reps[3];
                updateRep(reps, opt[l.cur].rep, l.lastSequence.off, l.lastSequence.litlen == 0);
                l.rep0 = reps[0];
                l.rep1 = reps[1];
                l.rep2 = reps[2];
            } else {
                l.rep0 = opt[l.cur].rep[0];
                l.rep1 = opt[l.cur].rep[1];
                l.rep2 = opt[l.cur].rep[2];
            }
            // ZSTD's reverse traversal copies the entries of the chosen path to the front of the table; here only their POSITIONS
            // are listed: bk[] (written with every finished entry) gives the way back, fw[] takes the list the store loop reads.
            // Both are 2 bytes per position: the walk stays inside a cache line or two instead of one 32-byte entry per step.
            // fw[top .. cur) = the positions of the path, in forward order (filled from the top: a path has at most cur entries)
            l.storeEnd = l.cur;
            U32 top = l.cur;
            for (U32 sp = l.cur; sp > 0;) {
                const U32 backDist = w.bk[sp];
                w.fw[--top] = (U16)sp;
                sp = (sp > backDist) ? sp - backDist : 0;
            }
            l.storePos = top;
            l.state = ST_STORE;
        } while (0);
        if (l.j == 0 && l.state == ST_DONE)
            sh.done = 1;
        ZS_GRP_END

        ZS_GRP_TICK(7);
        // ================= the group leaves the loop together =================
        bool fin = false;
        ZS_GRP_EACH(l)
        if (sh.done) {
            fin = true;
            if (l.j == 0) {
                rep[0] = l.rep0;
                rep[1] = l.rep1;
                rep[2] = l.rep2;
            }
        }
        ZS_GRP_END
        if (fin)
            break;
    }
#if defined(__HIP_DEVICE_COMPILE__) && defined(ZS_GRP_PROF)
    if (threadIdx.x == 0 && blockIdx.x % 400 == 7)
        printf("zsprof wave %u trips %llu cycles A %llu (store %llu next %llu first %llu) B %llu C %llu D %llu E %llu F %llu H %llu I %llu (price %llu chunkend %llu)\n", blockIdx.x,
               zs_prof[9], zs_prof[0] + zs_prof[10] + zs_prof[11], zs_prof[10], zs_prof[11], zs_prof[0], zs_prof[1], zs_prof[2], zs_prof[3], zs_prof[4], zs_prof[5],
               zs_prof[6], zs_prof[7] + zs_prof[12], zs_prof[12], zs_prof[7]);
#endif
    return srcSize - lanes[0].anchor;
}

} // namespace zs
