"""Multi-GPU `create` into ONE archive: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests).

Samples are dealt round-robin (sample i belongs to rank i mod N).  Every rank keeps the complete classification state --
splitters, the (k1,k2) -> group map, terminator lists and all group references in its own HBM -- so any rank can scan,
classify and LZ-encode any sample.  What orders the job is the reference's registration contract (group ids are minted
first-come in sample order, agc_compressor.cpp:954-1050): samples are therefore COMMITTED in order.  Per sample:

    owner:   prepare + commit (head)     scan + classification + speculative LZ-encode on its GPU ahead of its turn; at its turn
                                         the order-dependent registration and the HEAD of the commit record
    all:     broadcast(record head)      one collective per sample: length, then the bytes (uint8 tensor on the backend's device)
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
import time

import numpy as np


def concatenated_units(contig_names_per_file, pack_cardinality):
    """-c mode.  contig_names_per_file: for every input file, in command-line order, the names of its contigs.
    -> list of units; a unit = list of (file index, contig index) in order.  A contig whose name was seen before is skipped as the
    reference skips it ("already in the archive", agc_compressor.cpp:2201-2205); the last unit is what is left (possibly empty):
    the registration token the reference sends after the last file."""
    units, cur, seen = [], [], set()
    for fi, names in enumerate(contig_names_per_file):
        for ci, name in enumerate(names):
            if name in seen:
                continue
            seen.add(name)
            cur.append((fi, ci))
            if len(cur) >= pack_cardinality:
                units.append(cur)
                cur = []
    units.append(cur)
    return units


class DistCompressor:
    """wraps an agc_amd.host.Compressor that was given set_distributed(rank, world, writer) before create().
    device: this rank's GPU (torch.device) or None (CPU tests).  The collectives run on the GPU when the backend is nccl
    (RCCL: the record never leaves HBM on its way between GPUs) and on host tensors otherwise (gloo)."""

    def __init__(self, cmp_, dist, rank, world, device=None, writer=0):
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
        self._check_placement()
        self.warm_up()

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
        seconds, which would otherwise land in the first timed sample of that pair), an all_gather and a gather (Close)"""
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
        dist.all_gather([torch.zeros(8, dtype=torch.uint8, device=self.comm) for _ in range(self.world)], t)
        dist.gather(t, [torch.zeros(8, dtype=torch.uint8, device=self.comm) for _ in range(self.world)] if self.rank == self.writer else None,
                    dst=self.writer)
        if self.comm.type == "cuda":
            torch.cuda.synchronize(self.comm)

    def owner_of(self, i):
        return i % self.world

    def add_sample(self, sample_name=None, contig_names=None, d_codes=None, ctg_off=None):
        """SPMD: every rank calls this once per sample, in the same order; only the owner passes the data."""
        torch, dist = self.torch, self.dist
        i = self.next_sample
        self.next_sample += 1
        owner = self.owner_of(i)
        rec = body = None
        if self.rank == owner:
            t0 = time.perf_counter()
            self.cmp.add_sample_dev(sample_name, contig_names, d_codes, ctg_off)
            self.seconds["commit"] += time.perf_counter() - t0
            rec, body = self.cmp.last_record(copy=False), self.cmp.last_record_body(copy=False)
        self._publish(owner, rec, body)
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
        """Close() with the entropy stage of the delta packs spread over all ranks: the writer hands the pending packs out
        (one broadcast), rank r compresses a contiguous run of them -- the runs hold about the same number of bytes -- on its own
        GPU (agc_hip_zstd17_batch), the frames go back to the writer (gather), which finishes the archive.  No per-pack Python
        work anywhere (a human collection closes 50 k packs).
        zstd_raw(src uint8 array, off uint64[n + 1]) -> (frames uint8 array, foff uint64[n + 1]); default: this rank's GPU.
        Every rank must call this instead of Compressor.close()."""
        torch, dist = self.torch, self.dist
        default_raw = zstd_raw is None
        ctx = None
        if zstd_raw is None:
            from agc_amd import capi
            ctx = capi.Context.from_handle(self.cmp.hip_ctx())
            zstd_raw = ctx.zstd17_batch_raw
        writer = self.rank == self.writer
        if writer:
            src, off = self.cmp.close_collect_packs()
            meta = torch.tensor([off.size - 1, int(off[-1])], dtype=torch.int64, device=self.comm)
        else:
            meta = torch.zeros(2, dtype=torch.int64, device=self.comm)
        dist.broadcast(meta, src=self.writer)
        n, total = int(meta[0]), int(meta[1])
        if n == 0:
            if writer:
                self.cmp.close_provide_frames(np.zeros(0, np.uint8), np.zeros(1, np.uint64))
            self.cmp.close(n_threads)
            return
        d_off = torch.from_numpy(off.astype(np.int64)).to(self.comm) if writer else torch.zeros(n + 1, dtype=torch.int64, device=self.comm)
        d_src = torch.from_numpy(src).to(self.comm) if writer else torch.empty(total, dtype=torch.uint8, device=self.comm)
        dist.broadcast(d_off, src=self.writer)
        dist.broadcast(d_src, src=self.writer)   # (nccl: HBM -> HBM over xGMI)
        h_off = d_off.cpu().numpy().astype(np.uint64)
        # the packs stay where the broadcast put them: with a GPU the kernel reads them from HBM (RCCL delivered them there; under
        # gloo they are uploaded once) -- no HBM -> host -> HBM round trip of the inputs
        dev_path = self.hbm is not None and default_raw
        if dev_path:
            d_hbm = d_src if d_src.is_cuda else d_src.to(self.hbm)
            torch.cuda.synchronize(self.hbm)
        else:
            h_src = d_src.cpu().numpy()
        # rank r takes packs [cut[r], cut[r + 1]): equal shares of the bytes
        cut = np.searchsorted(h_off, (np.arange(self.world + 1, dtype=np.float64) * total / self.world).astype(np.uint64), side="left")
        cut[0], cut[-1] = 0, n
        cut = np.maximum.accumulate(np.minimum(cut, n))
        a, b = int(cut[self.rank]), int(cut[self.rank + 1])
        if b > a:
            if dev_path:
                frames, foff = ctx.zstd17_batch_raw_dev(d_hbm.data_ptr(), h_off[a:b + 1])
            else:
                frames, foff = zstd_raw(h_src[int(h_off[a]):int(h_off[b])], h_off[a:b + 1] - h_off[a])
            frames = np.ascontiguousarray(frames, dtype=np.uint8)
            sizes = np.diff(foff.astype(np.int64))
        else:
            frames, sizes = np.zeros(0, np.uint8), np.zeros(0, np.int64)
        # every rank's frame sizes (padded to the longest run), then the bytes (padded to the largest total)
        per = int(np.max(np.diff(cut))) if n else 0
        sz = torch.zeros(max(per, 1), dtype=torch.int64, device=self.comm)
        if sizes.size:
            sz[:sizes.size] = torch.from_numpy(sizes).to(self.comm)
        all_sz = [torch.zeros(max(per, 1), dtype=torch.int64, device=self.comm) for _ in range(self.world)]
        dist.all_gather(all_sz, sz)
        tot = [int(x.sum()) for x in all_sz]
        cap = max(max(tot), 1)
        buf = torch.zeros(cap, dtype=torch.uint8, device=self.comm)
        if frames.size:
            buf[:frames.size] = torch.from_numpy(frames).to(self.comm)
        gathered = [torch.zeros(cap, dtype=torch.uint8, device=self.comm) for _ in range(self.world)] if writer else None
        dist.gather(buf, gathered, dst=self.writer)
        if writer:
            sizes_all = np.concatenate([all_sz[r].cpu().numpy()[:int(cut[r + 1] - cut[r])] for r in range(self.world)]).astype(np.uint64)
            foff_all = np.zeros(n + 1, np.uint64)
            foff_all[1:] = np.cumsum(sizes_all)
            out = np.concatenate([gathered[r].cpu().numpy()[:tot[r]] for r in range(self.world)]) if n else np.zeros(0, np.uint8)
            self.cmp.close_provide_frames(out, foff_all)
        self.cmp.close(n_threads)

    def _publish(self, owner, rec, body):
        """one sample's commit record: the head to every rank (broadcast), the delta body to the writer only (point to point);
        ranks other than the owner apply it.  rec: the owner's head (numpy uint8 view into its compressor), None elsewhere.
        body: the owner's body, or a function that finishes the commit and returns it -- called AFTER the head is out, so the
        other ranks go on while the owner indexes its new references, encodes what is left and builds the body."""
        torch, dist = self.torch, self.dist
        t0 = time.perf_counter()
        n = torch.zeros(1, dtype=torch.int64, device=self.comm)
        if rec is not None:
            n[0] = rec.size
        dist.broadcast(n, src=owner)
        size = int(n[0])
        buf = torch.from_numpy(rec).to(self.comm) if rec is not None else torch.empty(size, dtype=torch.uint8, device=self.comm)
        dist.broadcast(buf, src=owner)
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
            # the head is parsed on the host; the new references are registered from the copy in this rank's HBM when there is one
            host = np.ascontiguousarray(buf.cpu().numpy())
            d_buf = buf if buf.is_cuda else (buf.to(self.hbm) if self.hbm is not None else None)
            if d_buf is not None:
                torch.cuda.synchronize(self.hbm)
            self.cmp.apply_record(host.ctypes.data, size, d_buf.data_ptr() if d_buf is not None else None,
                                  b_view.ctypes.data if b_view is not None else None, bsize if b_view is not None else 0)
        t3 = time.perf_counter()
        self.seconds["head"] += t1 - t0
        self.seconds["body"] += t2 - t1
        self.seconds["apply"] += t3 - t2

    def _commit_and_publish(self, i):
        t0 = time.perf_counter()
        self.cmp.commit_prepared_head()
        self.seconds["commit"] += time.perf_counter() - t0
        self._publish(self.rank, self.cmp.last_record(copy=False), self._finish_commit)
        self.next_sample = i + 1

    def _finish_commit(self):
        self.cmp.commit_prepared_finish()
        return self.cmp.last_record_body(copy=False)

    def _receive(self, owner):
        self._publish(owner, None, None)
        self.next_sample += 1
