"""Multi-GPU `create` into ONE archive: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests).

Samples are dealt round-robin (sample i belongs to rank i mod N).  Every rank keeps the complete classification state --
splitters, the (k1,k2) -> group map, terminator lists and all group references in its own HBM -- so any rank can scan,
classify and LZ-encode any sample.  What orders the job is the reference's registration contract (group ids are minted
first-come in sample order, agc_compressor.cpp:954-1050): samples are therefore COMMITTED in order.  Per sample:

    owner:   prepare + commit (head)     scan + classification + speculative LZ-encode on its GPU ahead of its turn; at its turn
                                         the order-dependent registration and the HEAD of the commit record
    all:     broadcast(record head)      ONE collective per sample: a 64-byte message header (sizes) + the head, in a buffer of a
                                         capacity every rank derives from the sizes seen so far (a head that does not fit announces
                                         its size in the header and its rest follows in a second broadcast: the reference sample)
    owner:   commit (finish)             new references indexed on its GPU, leftover deltas, the record's BODY -- after the head is out
    owner -> writer: send(record body)   the LZ deltas, point to point: only the writer needs them
    others:  apply_record(head)          same group ids / map / terminator updates; the newly minted reference segments inside the
                                         head are registered in this rank's HBM straight from the broadcast buffer
    writer:  apply_record(head, body)    the same + bookkeeping, zstd parts, collection metadata, archive

The head carries the symbols of every new reference (the "all-gather of newly-minted reference segments" of the north
star -- a broadcast from the minting rank, since exactly one rank mints at a time) and the raw segments: ~3 MB per human-size
sample; the body the deltas (~16 B per SNP): ~22 MB.  The archive is byte-identical to the single-GPU / reference one.

-c (concatenated genomes): the unit that is dealt round-robin is the reference's own registration unit -- a run of pack_cardinality
contigs across the input files, each contig a sample of its own -- followed by the empty registration the reference always sends at
the end (agc_compressor.cpp:2155-2238); concatenated_units() below cuts them.
append: every rank loads the input archive (host.Compressor.append); compress(prefetch=False) -- a packed group answers Estimate with
0 until a record unpacks it, on every rank, so samples are prepared at their turn.  Not covered: append together with -c.
"""
import os
import time

import numpy as np


def concatenated_units(contig_names_per_file, pack_cardinality, already=0, seen=()):
    """-c mode.  contig_names_per_file: for every input file, in command-line order, the names of its contigs.
    -> list of units; a unit = list of (file index, contig index) in order.  A contig whose name was seen before is skipped as the
    reference skips it ("already in the archive", agc_compressor.cpp:2201-2205); the last unit is what is left (possibly empty):
    the registration token the reference sends after the last file.
    append: `already` = samples of the input archive modulo the pack cardinality -- the first unit completes the batch the archive
    ended in (agc_compressor.cpp:2150-2153: processed_samples starts at the collection's sample count) -- and `seen` = the contig
    names it holds (archive_contig_names)."""
    units, cur, seen = [], [], set(seen)
    fill = int(already) % pack_cardinality
    for fi, names in enumerate(contig_names_per_file):
        for ci, name in enumerate(names):
            if name in seen:
                continue
            seen.add(name)
            cur.append((fi, ci))
            fill += 1
            if fill >= pack_cardinality:
                units.append(cur)
                cur, fill = [], 0
    units.append(cur)
    return units


def archive_contig_names(path):
    """(number of samples, names of all contigs, pack cardinality) of an archive: what `append -c` continues from
    (concatenated_units)"""
    from . import reader
    f = reader.CAGCFile()
    if not f.Open(path, False):
        raise RuntimeError("cannot open " + path)
    samples = f.ListSample()
    names = [c for s in samples for c in f.ListCtg(s)]
    b = (f.GetParams() or {}).get("pack_cardinality")
    f.Close()
    return len(samples), names, b


class DistCompressor:
    """wraps an agc_amd.host.Compressor that was given set_distributed(rank, world, writer) before create().
    device: this rank's GPU (torch.device) or None (CPU tests).  The collectives run on the GPU when the backend is nccl
    (RCCL: the record never leaves HBM on its way between GPUs) and on host tensors otherwise (gloo)."""

    def __init__(self, cmp_, dist, rank, world, device=None, writer=0, zstd_raw=None):
        import torch
        self.torch = torch
        self.cmp = cmp_
        self.dist = dist
        self.rank, self.world, self.writer = rank, world, writer
        self.hbm = device if device is not None and device.type == "cuda" else None
        self.comm = self.hbm if (self.hbm is not None and dist.get_backend() == "nccl") else torch.device("cpu")
        self.next_sample = 0
        self.bytes_broadcast = 0  # record heads, to every rank
        self.bytes_p2p = 0        # record bodies (deltas), owner -> writer
        # host seconds of this rank per stage: prepare (scan + classification + speculative encode of an own sample), commit (its
        # order-dependent half up to the record's head), head (head size + head broadcast), finish (the owner's rest: new references
        # indexed, leftover deltas, the body), body (finish + delta body to the writer), apply (the other ranks' records applied here)
        self.seconds = {"prepare": 0.0, "commit": 0.0, "head": 0.0, "finish": 0.0, "body": 0.0, "apply": 0.0}
        self.n_records = 0          # samples published (sample 0, the reference, included)
        self.bytes_head_last = 0    # head size of the last record (a typical sample's: the reference sample's is the collection)
        self.n_collectives = 0      # broadcasts spent on record heads
        # transport buffers of the record heads, reused for every sample: the message = 64-byte header + head.  Under nccl the
        # broadcast runs HBM -> HBM (`_dmsg`), the owner fills it from the compressor's pinned record with one async copy and the other
        # ranks read header + head back into a pinned host buffer (`_hmsg`) -- no pageable staging, no allocation per sample; under gloo
        # the broadcast runs on `_hmsg` itself
        import os
        # (AGC_AMD_DIST_MSG_CAP0 / _MAX: the tests shrink the message so that the second broadcast of a long head and the capacity's
        # adaptation are walked through by small collections)
        self.MSG_CAP0 = max(self.MSG_HDR + 8, int(os.environ.get("AGC_AMD_DIST_MSG_CAP0", self.MSG_CAP0)))
        self.MSG_CAP_MAX = max(self.MSG_CAP0, int(os.environ.get("AGC_AMD_DIST_MSG_CAP_MAX", self.MSG_CAP_MAX)))
        self._cap = self.MSG_CAP0
        self._prev_msg_bytes = 0
        self._hmsg = self._host_buffer(self._cap)
        self._dmsg = torch.empty(self._cap, dtype=torch.uint8, device=self.comm) if self.comm.type == "cuda" else None
        self._dapply = None         # (gloo with a GPU: the head's copy in this rank's HBM the new references are registered from)
        # ---- full delta packs dealt to the ranks' entropy stages IN THE MIDDLE of the run (_deal_step) ----
        # AGC_AMD_DEAL_MIN_MB: the writer deals once that much has piled up (default 64; negative: only Close deals);
        # AGC_AMD_DEAL_EVERY: a control step (one small all_gather) every that many samples (default 4)
        import threading
        import queue
        self.deal_min = int(float(os.environ.get("AGC_AMD_DEAL_MIN_MB", "64")) * (1 << 20))
        self.deal_every = max(1, int(os.environ.get("AGC_AMD_DEAL_EVERY", "4")))
        self._zstd_raw = zstd_raw           # (src uint8 array, off uint64[n + 1]) -> (frames, foff); None: this rank's GPU
        self._deal_q = queue.Queue()        # shares this rank has to code: (deal id, packs tensor / array, offsets)
        self._deal_done = []                # ... and has coded: (deal id, frame sizes int64[n], frames uint8 array), oldest first
        self._deal_lock = threading.Lock()
        self._deal_thread = None
        self._deal_error = None
        self._deals_out = {}                # writer: (deal id, rank) -> (first pack, count) of the shares that are out
        self._since_deal = 0
        self.n_deals = 0                    # deals started (every rank counts the same)
        self.bytes_dealt = 0                # pack bytes that left the writer in deals
        self.seconds["deal"] = 0.0          # host seconds of this rank in control steps and transfers of deals
        self._check_placement()
        self.warm_up()

    MSG_CAP0 = 4 << 20      # capacity of the head message before any head has been seen
    MSG_CAP_MAX = 64 << 20  # ... and the most a typical head may claim (a larger one takes the second broadcast)
    MSG_HDR = 64            # bytes of the message header in front of the head (Impl::DIST_FRAME)

    def _host_buffer(self, n):
        torch = self.torch
        try:
            return torch.empty(n, dtype=torch.uint8, pin_memory=self.hbm is not None)
        except RuntimeError:
            return torch.empty(n, dtype=torch.uint8)

    def _next_cap(self, msg_bytes):
        """capacity of the NEXT head message, from the size of this one: every rank sees every size, so every rank derives the same"""
        # (heads of consecutive samples differ by ten or twenty percent -- a sample mints 40 to 90 groups: a quarter above the larger
        # of the last two keeps the second broadcast for the reference sample)
        big = max(msg_bytes, self._prev_msg_bytes)
        self._prev_msg_bytes = msg_bytes
        want = (big * 5 // 4 + (1 << 18) - 1) >> 18 << 18
        return min(self.MSG_CAP_MAX, max(self.MSG_CAP0, want))

    def _set_cap(self, cap):
        if cap == self._cap:
            return
        self._cap = cap
        if self._hmsg.numel() < cap:
            self._hmsg = self._host_buffer(cap)
        if self._dmsg is not None and self._dmsg.numel() < cap:
            self._dmsg = self.torch.empty(cap, dtype=self.torch.uint8, device=self.comm)

    def _check_placement(self):
        """one process per GPU: a world larger than the visible devices (or two ranks on one device) is a launch mistake that RCCL
        reports as a hang or an obscure error much later -- said here, clearly, before the first collective"""
        if self.hbm is None or self.dist.get_backend() != "nccl":
            return
        n_dev = self.torch.cuda.device_count()
        if self.world > n_dev:
            raise RuntimeError(f"{self.world} ranks but only {n_dev} visible GPU(s): the RCCL mode runs one process per GPU "
                               f"(AGC_BENCH_ONE_GPU=1 with the gloo backend shares one device for functional runs)")
        mine = self.torch.tensor([self.hbm.index if self.hbm.index is not None else self.torch.cuda.current_device()], dtype=self.torch.int64, device=self.comm)
        every = [self.torch.zeros(1, dtype=self.torch.int64, device=self.comm) for _ in range(self.world)]
        self.dist.all_gather(every, mine)
        devs = [int(x[0]) for x in every]
        if len(set(devs)) != len(devs):
            raise RuntimeError(f"ranks share a GPU under the nccl backend: devices per rank {devs}")

    def warm_up(self):
        """every communication pattern of the run once, with a few bytes: a broadcast from every rank (the record heads), a
        point-to-point message from every rank to the writer (the record bodies; RCCL sets a pair's channel up on its first use --
        seconds, which would otherwise land in the first timed sample of that pair) and one back (Close: the packs)"""
        torch, dist = self.torch, self.dist
        t = torch.zeros(8, dtype=torch.uint8, device=self.comm)
        for src in range(self.world):
            dist.broadcast(t, src=src)
        for r in range(self.world):
            if r == self.writer:
                continue
            if self.rank == r:
                dist.send(t, dst=self.writer)
            elif self.rank == self.writer:
                dist.recv(t, src=r)
        for r in range(self.world):  # ... and back: Close hands every rank its share of the packs point to point
            if r == self.writer:
                continue
            if self.rank == self.writer:
                dist.send(t, dst=r)
            elif self.rank == r:
                dist.recv(t, src=self.writer)
        if self.comm.type == "cuda":
            torch.cuda.synchronize(self.comm)

    def owner_of(self, i):
        return i % self.world

    # ---- deals: the reference's workers code a delta pack the moment it is full while the others go on (segment.cpp:34-80,
    # segment.h:258-280); here the full packs pile up at the writer (their parts hold their places in the archive) and are DEALT to
    # every rank's entropy stage a few samples later: a rank codes its share on a thread of its own, on its own GPU's entropy stream,
    # beside the samples it goes on preparing, and the frames travel back at a later control step.  Close then has the packs still
    # open left to deal, not every pack of the run.
    def _code_share(self, packs, off):
        if self._zstd_raw is not None:
            src = packs.cpu().numpy() if hasattr(packs, "cpu") else packs
            return self._zstd_raw(src, off)
        from agc_amd import capi
        ctx = capi.Context.from_handle(self.cmp.hip_ctx())
        if self.hbm is not None:
            d = packs if (hasattr(packs, "is_cuda") and packs.is_cuda) else self.torch.from_numpy(np.ascontiguousarray(packs)).to(self.hbm)
            self.torch.cuda.synchronize(self.hbm)
            return ctx.zstd17_batch_raw_dev(d.data_ptr(), off)
        return ctx.zstd17_batch_raw(packs.numpy() if hasattr(packs, "numpy") else packs, off)

    def _deal_main(self):
        while True:
            job = self._deal_q.get()
            if job is None:
                return
            deal, packs, off = job
            try:
                frames, foff = self._code_share(packs, off)
                with self._deal_lock:
                    self._deal_done.append((deal, np.diff(foff.astype(np.int64)), np.ascontiguousarray(frames, dtype=np.uint8)))
            except Exception as e:  # noqa: BLE001 -- reported at the next control step, on the thread that drives the ranks
                self._deal_error = e
            finally:
                self._deal_q.task_done()

    def _deal_step(self, final=False):
        """SPMD, at the same sample boundaries on every rank: ONE small all_gather says whether the writer starts a deal (and every
        rank's share of it) and which rank has a finished share to return; the transfers follow point to point."""
        torch, dist = self.torch, self.dist
        t0 = time.perf_counter()
        W, writer = self.world, self.rank == self.writer
        if self._deal_error is not None:
            raise self._deal_error
        v = np.zeros(2 + 2 * W, np.int64)
        with self._deal_lock:
            if self._deal_done and not writer:
                v[0] = self._deal_done[0][0]
        src = off = cut = None
        if writer and not final and self.deal_min >= 0 and self.cmp.deferred_pack_bytes() >= max(self.deal_min, 1):
            deal, src, off = self.cmp.deal_collect_packs()
            if deal:
                n, total = off.size - 1, int(off[-1])
                cut = np.searchsorted(off, (np.arange(W + 1, dtype=np.float64) * total / W).astype(np.uint64), side="left")
                cut[0], cut[-1] = 0, n
                cut = np.maximum.accumulate(np.minimum(cut, n)).astype(np.int64)
                v[1] = deal
                for r in range(W):
                    v[2 + 2 * r], v[3 + 2 * r] = cut[r + 1] - cut[r], int(off[cut[r + 1]]) - int(off[cut[r]])
        mine = torch.from_numpy(v).to(self.comm)
        every = [torch.zeros(2 + 2 * W, dtype=torch.int64, device=self.comm) for _ in range(W)]
        dist.all_gather(every, mine)
        every = [e.cpu().numpy() for e in every]
        moved = False
        # ---- a new deal: every rank's share leaves the writer
        deal = int(every[self.writer][1])
        if deal:
            moved = True
            self.n_deals += 1
            plan = every[self.writer][2:].reshape(W, 2)
            if writer:
                sends = []
                for r in range(W):
                    a_, b_ = int(cut[r]), int(cut[r + 1])
                    if b_ == a_:
                        continue
                    if r == self.writer:
                        self.cmp.deal_keep_own(deal, a_, b_ - a_)  # (this rank's own entropy stage, beside its steps)
                        continue
                    self._deals_out[(deal, r)] = (a_, b_ - a_)
                    sends.append(dist.isend(torch.from_numpy((off[a_:b_ + 1] - off[a_]).astype(np.int64)).to(self.comm), dst=r))
                    if int(off[b_]) > int(off[a_]):
                        sends.append(dist.isend(torch.from_numpy(src[int(off[a_]):int(off[b_])]).to(self.comm), dst=r))
                    self.bytes_dealt += int(off[b_]) - int(off[a_])
                for w_ in sends:
                    w_.wait()
            elif plan[self.rank, 0]:
                n_, bytes_ = int(plan[self.rank, 0]), int(plan[self.rank, 1])
                d_o = torch.empty(n_ + 1, dtype=torch.int64, device=self.comm)
                d_x = torch.empty(bytes_, dtype=torch.uint8, device=self.comm)
                dist.recv(d_o, src=self.writer)
                if bytes_:
                    dist.recv(d_x, src=self.writer)
                if self._deal_thread is None:
                    import threading
                    self._deal_thread = threading.Thread(target=self._deal_main, daemon=True)
                    self._deal_thread.start()
                self._deal_q.put((deal, d_x, d_o.cpu().numpy().astype(np.uint64)))
        # ---- finished shares travel back (one per rank and control step, oldest first)
        for r in range(W):
            back = int(every[r][0])
            if not back or r == self.writer:
                continue
            moved = True
            if writer:
                first, count = self._deals_out.pop((back, r))
                d_s = torch.empty(count, dtype=torch.int64, device=self.comm)
                dist.recv(d_s, src=r)
                sz = d_s.cpu().numpy()
                d_f = torch.empty(int(sz.sum()), dtype=torch.uint8, device=self.comm)
                if d_f.numel():
                    dist.recv(d_f, src=r)
                foff = np.zeros(count + 1, np.uint64)
                foff[1:] = np.cumsum(sz)
                self.cmp.deal_provide_frames(back, first, count, d_f.cpu().numpy(), foff)
            elif r == self.rank:
                with self._deal_lock:
                    _d, sz, frames = self._deal_done.pop(0)
                dist.send(torch.from_numpy(sz).to(self.comm), dst=self.writer)
                if frames.size:
                    dist.send(torch.from_numpy(frames).to(self.comm), dst=self.writer)
        self.seconds["deal"] += time.perf_counter() - t0
        return moved

    def _after_sample(self):
        self._since_deal += 1
        if self.world > 1 and self._since_deal >= self.deal_every:
            self._since_deal = 0
            self._deal_step()

    def _settle_deals(self):
        """Close: every share that is out comes home first (each rank waits for its own coding thread, then control steps until none moves)"""
        if self.world <= 1:
            return
        self._deal_q.join()
        while True:
            moved = self._deal_step(final=True)
            flag = self.torch.tensor([1 if moved else 0], dtype=self.torch.int64, device=self.comm)
            self.dist.all_reduce(flag, op=self.dist.ReduceOp.MAX)
            if not int(flag[0]):
                break
        if self._deal_thread is not None:
            self._deal_q.put(None)
            self._deal_thread.join()
            self._deal_thread = None
        if self._deal_error is not None:
            raise self._deal_error

    def add_sample(self, sample_name=None, contig_names=None, d_codes=None, ctg_off=None):
        """SPMD: every rank calls this once per sample, in the same order; only the owner passes the data."""
        torch, dist = self.torch, self.dist
        i = self.next_sample
        self.next_sample += 1
        owner = self.owner_of(i)
        rec = body = None
        if self.rank == owner:
            t0 = time.perf_counter()
            try:
                self.cmp.add_sample_dev(sample_name, contig_names, d_codes, ctg_off)
            except Exception:
                self._announce_failure()  # (the other ranks are waiting in the broadcast of this sample's record)
                raise
            self.seconds["commit"] += time.perf_counter() - t0
            rec, body = self.cmp.last_record_framed(), self.cmp.last_record_body(copy=False)
        self._publish(owner, rec, body)
        self._after_sample()
        return owner

    def compress(self, n_total, get_sample, prefetch=True, start=0):
        """The samples start .. n_total-1, in order (sample indices are global: i belongs to rank i mod N).
        get_sample(i) -> (sample_name, contig_names, d_codes_ptr, ctg_off) is called for the samples
        this rank owns only; whatever backs d_codes_ptr must stay alive and unchanged until that sample is committed.
        prefetch: a rank classifies and speculatively encodes its NEXT sample (PrepareSampleDevice) before it joins the broadcasts of
        the samples in front of it, so the GPUs work in parallel and only the short commit (revalidation of the decisions that read
        state changed meanwhile + registration + record) is serial.  Adaptive mode (new splitters change later scans): the prepare
        ahead of the turn does not extend the splitter set; a sample that would have to, or whose set grew while it waited, is
        prepared again at its turn (CommitPreparedHead) -- collections whose samples rarely bring new splitters keep the overlap."""
        nxt = start + (self.rank - start) % self.world
        nxt = nxt if nxt < n_total else None
        prepared = None
        for i in range(start, n_total):
            owner = self.owner_of(i)
            if prefetch and prepared is None and nxt is not None:
                self._prepare(get_sample(nxt))
                prepared = nxt
            if self.rank == owner:
                if prepared is None:
                    self._prepare(get_sample(i))
                    prepared = i
                assert prepared == i
                self._commit_and_publish(i)
                prepared = None
                nxt = i + self.world if i + self.world < n_total else None
            else:
                self._receive(owner)
            self._after_sample()

    def _prepare(self, sample):
        """sample = (name, contig names, d_codes pointer | agc_amd.capi.Packed, ctg_off)"""
        name, names, data, off = sample
        t0 = time.perf_counter()
        if isinstance(data, int) or data is None:
            self.cmp.prepare_sample_dev(name, names, data, off)
        else:
            self.cmp.prepare_sample_packed_dev(name, names, data, off)
        self.seconds["prepare"] += time.perf_counter() - t0

    def close(self, zstd_raw=None, n_threads=8):
        """Close() with the entropy stage of the delta packs spread over all ranks: the writer hands every rank ITS run of the pending
        packs (point to point; the runs hold about the same number of bytes), rank r compresses it on its own GPU
        (agc_hip_zstd17_batch_dev), the frames go back to the writer as long as they are, which finishes the archive.  No per-pack Python
        work anywhere (a human collection closes 50 k packs).
        zstd_raw(src uint8 array, off uint64[n + 1]) -> (frames uint8 array, foff uint64[n + 1]); default: this rank's GPU.
        Every rank must call this instead of Compressor.close()."""
        torch, dist = self.torch, self.dist
        if zstd_raw is not None and self._zstd_raw is None:
            self._zstd_raw = zstd_raw
        self._settle_deals()
        default_raw = zstd_raw is None
        ctx = None
        if zstd_raw is None:
            from agc_amd import capi
            ctx = capi.Context.from_handle(self.cmp.hip_ctx())
            zstd_raw = ctx.zstd17_batch_raw
        writer = self.rank == self.writer
        W = self.world
        src = off = None
        if writer:
            src, off = self.cmp.close_collect_packs()
            n, total = off.size - 1, int(off[-1])
            # rank r takes packs [cut[r], cut[r + 1]): equal shares of the bytes
            cut = np.searchsorted(off, (np.arange(W + 1, dtype=np.float64) * total / W).astype(np.uint64), side="left")
            cut[0], cut[-1] = 0, n
            cut = np.maximum.accumulate(np.minimum(cut, n)).astype(np.int64)
            plan = np.zeros((W, 2), np.int64)  # per rank: packs, bytes
            for r in range(W):
                plan[r] = (cut[r + 1] - cut[r], int(off[cut[r + 1]]) - int(off[cut[r]]))
            d_plan = torch.from_numpy(plan.reshape(-1)).to(self.comm)
        else:
            d_plan = torch.zeros(2 * W, dtype=torch.int64, device=self.comm)
        dist.broadcast(d_plan, src=self.writer)  # (one small collective: who gets how much)
        plan = d_plan.cpu().numpy().reshape(W, 2)
        if int(plan[:, 0].sum()) == 0:
            if writer:
                self.cmp.close_provide_frames(np.zeros(0, np.uint8), np.zeros(1, np.uint64))
            self.cmp.close(n_threads)
            return
        my_n, my_bytes = int(plan[self.rank, 0]), int(plan[self.rank, 1])
        # ---- the packs: every rank receives ITS OWN byte range only, point to point (round 4 broadcast all of them to all ranks:
        # 4.4 GB at N = 8 where a rank needs an eighth); the writer's share stays where it is
        dev_path = self.hbm is not None and default_raw
        if writer:
            sends = []
            for r in range(W):
                if r == self.writer or plan[r, 0] == 0:
                    continue
                a_, b_ = int(cut[r]), int(cut[r + 1])
                o_ = torch.from_numpy((off[a_:b_ + 1] - off[a_]).astype(np.int64)).to(self.comm)
                x_ = torch.from_numpy(src[int(off[a_]):int(off[b_])]).to(self.comm)
                sends.append(dist.isend(o_, dst=r))
                if x_.numel():  # (packs that are all empty: nothing but their offsets travels, as on the frames leg)
                    sends.append(dist.isend(x_, dst=r))
            a_, b_ = int(cut[self.rank]), int(cut[self.rank + 1])
            my_off = (off[a_:b_ + 1] - off[a_]).astype(np.uint64)
            my_src = src[int(off[a_]):int(off[b_])]
            d_mine = torch.from_numpy(my_src).to(self.hbm) if (dev_path and my_n) else None
        elif my_n:
            d_o = torch.empty(my_n + 1, dtype=torch.int64, device=self.comm)
            d_x = torch.empty(my_bytes, dtype=torch.uint8, device=self.comm)
            dist.recv(d_o, src=self.writer)
            if my_bytes:
                dist.recv(d_x, src=self.writer)   # (nccl: HBM -> HBM over xGMI, and the kernel reads the packs where they landed)
            my_off = d_o.cpu().numpy().astype(np.uint64)
            if dev_path:
                d_mine = d_x if d_x.is_cuda else d_x.to(self.hbm)
            else:
                my_src = d_x.cpu().numpy()
        if my_n:
            if dev_path:
                torch.cuda.synchronize(self.hbm)
                frames, foff = ctx.zstd17_batch_raw_dev(d_mine.data_ptr(), my_off)
            else:
                frames, foff = zstd_raw(my_src, my_off)
            frames = np.ascontiguousarray(frames, dtype=np.uint8)
            sizes = np.diff(foff.astype(np.int64))
        else:
            frames, sizes = np.zeros(0, np.uint8), np.zeros(0, np.int64)
        if writer:
            for w_ in sends:  # (the other ranks' packs left while this rank's own share was being compressed)
                w_.wait()
        # ---- the frames back to the writer, as long as they are (sizes first: the writer knows every rank's pack count)
        if writer:
            all_sizes, all_frames = [None] * W, [None] * W
            all_sizes[self.rank], all_frames[self.rank] = sizes, frames
            for r in range(W):
                if r == self.writer or plan[r, 0] == 0:
                    continue
                d_s = torch.empty(int(plan[r, 0]), dtype=torch.int64, device=self.comm)
                dist.recv(d_s, src=r)
                s_ = d_s.cpu().numpy()
                d_f = torch.empty(int(s_.sum()), dtype=torch.uint8, device=self.comm)
                if d_f.numel():
                    dist.recv(d_f, src=r)
                all_sizes[r], all_frames[r] = s_, d_f.cpu().numpy()
            order = [r for r in range(W) if plan[r, 0]]
            sizes_all = np.concatenate([all_sizes[r] for r in order]).astype(np.uint64)
            foff_all = np.zeros(sizes_all.size + 1, np.uint64)
            foff_all[1:] = np.cumsum(sizes_all)
            out = np.concatenate([all_frames[r] for r in order])
            self.cmp.close_provide_frames(out, foff_all)
        elif my_n:
            dist.send(torch.from_numpy(sizes).to(self.comm), dst=self.writer)
            if frames.size:
                dist.send(torch.from_numpy(frames).to(self.comm), dst=self.writer)
        self.cmp.close(n_threads)

    def _announce_failure(self):
        """the owner of the sample whose record everybody waits for could not make it: a header that says so goes out in the
        record's place, so that the other ranks raise instead of waiting in the broadcast until its timeout"""
        torch, dist = self.torch, self.dist
        H, cap = self.MSG_HDR, self._cap
        msg = self._dmsg if self._dmsg is not None else self._hmsg
        hdr = np.zeros(H, np.uint8)
        hdr.view(np.uint64)[0] = 0x58434741  # "AGCX"
        msg[:H].copy_(torch.from_numpy(hdr))
        try:
            dist.broadcast(msg[:cap], src=self.rank)
        except Exception:
            pass  # (the failure that brought us here is the one to report)

    def _publish(self, owner, rec, body):
        """one sample's commit record: the head to every rank (one broadcast), the delta body to the writer only (point to point);
        ranks other than the owner apply it.  rec: the owner's head as the compressor frames it (numpy uint8 view of its pinned
        buffer: 64 bytes for the message header + the head), None elsewhere.
        body: the owner's body, or a function that finishes the commit and returns it -- called AFTER the head is out, so the
        other ranks go on while the owner indexes its new references, encodes what is left and builds the body."""
        torch, dist = self.torch, self.dist
        t0 = time.perf_counter()
        H, cap = self.MSG_HDR, self._cap
        on_dev = self._dmsg is not None
        msg = self._dmsg if on_dev else self._hmsg
        if rec is not None:
            if rec.size < H:
                self._announce_failure()
                raise RuntimeError("the compressor has no commit record for this sample")
            hdr = rec[:H].view(np.uint64)
            hdr[:] = 0
            hdr[0], hdr[1] = 0x4D434741, rec.size - H  # "AGCM", bytes of the head
            first = min(rec.size, cap)
            msg[:first].copy_(torch.from_numpy(rec[:first]), non_blocking=True)  # (pinned -> HBM: one async copy; gloo: a memcpy)
        dist.broadcast(msg[:cap], src=owner)
        self.n_collectives += 1
        hm = self._hmsg
        if rec is None:
            if on_dev:
                # header + what a head of the usual size needs, in one copy and one wait; the rest only if this head is longer
                guess = min(cap, max(H, ((self.bytes_head_last + H) * 9 // 8 + 4095) & ~4095))
                hm[:guess].copy_(msg[:guess], non_blocking=True)
                torch.cuda.current_stream(self.comm).synchronize()
            h = hm[:H].numpy().view(np.uint64)
            if int(h[0]) == 0x58434741:  # "AGCX": _announce_failure
                raise RuntimeError(f"rank {self.rank}: rank {owner} failed to commit its sample (its own message says why)")
            if int(h[0]) != 0x4D434741:
                raise RuntimeError(f"rank {self.rank}: bad record message from rank {owner}")
            size = int(h[1])
            if on_dev and min(size + H, cap) > guess:
                hm[guess:min(size + H, cap)].copy_(msg[guess:min(size + H, cap)], non_blocking=True)
                torch.cuda.current_stream(self.comm).synchronize()
        else:
            size = rec.size - H
        rest = size + H - cap
        big = None
        if rest > 0:
            # a head beyond the message's capacity (the reference sample: every reference segment of the collection): the rest in a
            # second broadcast, sized by the header everybody has by now
            if rec is not None:
                big = torch.from_numpy(rec[cap:]).to(self.comm)
            else:
                big = torch.empty(rest, dtype=torch.uint8, device=self.comm)
            dist.broadcast(big, src=owner)
            self.n_collectives += 1
        self._set_cap(self._next_cap(min(size + H, self.MSG_CAP_MAX)))
        self.n_records += 1
        self.bytes_head_last = size
        self.bytes_broadcast += size
        t1 = time.perf_counter()
        b_view, bsize = None, 0
        if self.rank == owner:
            failure = None
            if callable(body):
                tf = time.perf_counter()
                try:
                    body = body()
                except Exception as e:  # the writer waits for a size: it gets one that says "the owner failed", then everybody stops
                    failure, body = e, None
                self.seconds["finish"] += time.perf_counter() - tf
            if owner != self.writer:
                bsize = int(body.size) if failure is None else -1
                dist.send(torch.tensor([bsize], dtype=torch.int64, device=self.comm), dst=self.writer)
                if bsize > 0:
                    dist.send(torch.from_numpy(body).to(self.comm), dst=self.writer)
            if failure is not None:
                raise failure
        elif self.rank == self.writer:
            nb = torch.zeros(1, dtype=torch.int64, device=self.comm)
            dist.recv(nb, src=owner)
            bsize = int(nb[0])
            if bsize < 0:
                raise RuntimeError(f"rank {owner} failed while finishing the commit of sample {self.next_sample - 1} (its message is on its stderr)")
            if bsize:
                # straight into the pinned buffer the bookkeeping will read (no staging copy on the host)
                b_view = self.cmp.record_body_buffer(bsize)
                if self.comm.type == "cpu":
                    dist.recv(torch.from_numpy(b_view), src=owner)
                else:
                    bt = torch.empty(bsize, dtype=torch.uint8, device=self.comm)
                    dist.recv(bt, src=owner)
                    torch.from_numpy(b_view).copy_(bt)
        self.bytes_p2p += bsize
        t2 = time.perf_counter()
        if self.rank != owner:
            # the head is parsed where it lies in pinned host memory; the new references are registered from its copy in this rank's
            # HBM (the broadcast buffer under nccl; uploaded once under gloo with a GPU)
            if big is None:
                host = hm[H:H + size].numpy()
                d_ptr = None
                if on_dev:
                    d_ptr = msg.data_ptr() + H
                elif self.hbm is not None:
                    if self._dapply is None or self._dapply.numel() < size:
                        self._dapply = torch.empty(max(size * 5 // 4, 1 << 20), dtype=torch.uint8, device=self.hbm)
                    self._dapply[:size].copy_(hm[H:H + size], non_blocking=True)
                    torch.cuda.synchronize(self.hbm)
                    d_ptr = self._dapply.data_ptr()
            else:
                # (the two pieces side by side, once per archive)
                whole = torch.empty(size, dtype=torch.uint8, device=self.comm)
                whole[:cap - H].copy_(msg[H:cap])
                whole[cap - H:].copy_(big)
                host = np.ascontiguousarray(whole.cpu().numpy())
                d_whole = whole if whole.is_cuda else (whole.to(self.hbm) if self.hbm is not None else None)
                if d_whole is not None:
                    torch.cuda.synchronize(self.hbm)
                d_ptr = d_whole.data_ptr() if d_whole is not None else None
            self.cmp.apply_record(host.ctypes.data, size, d_ptr, b_view.ctypes.data if b_view is not None else None, bsize if b_view is not None else 0)
        t3 = time.perf_counter()
        self.seconds["head"] += t1 - t0
        self.seconds["body"] += t2 - t1
        self.seconds["apply"] += t3 - t2

    def _commit_and_publish(self, i):
        t0 = time.perf_counter()
        try:
            self.cmp.commit_prepared_head()
        except Exception:
            self._announce_failure()
            raise
        self.seconds["commit"] += time.perf_counter() - t0
        self._publish(self.rank, self.cmp.last_record_framed(), self._finish_commit)
        self.next_sample = i + 1

    def _finish_commit(self):
        self.cmp.commit_prepared_finish()
        return self.cmp.last_record_body(copy=False)

    def _receive(self, owner):
        self._publish(owner, None, None)
        self.next_sample += 1
