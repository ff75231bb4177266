"""`agc create` / `agc append` on N GPUs into ONE archive (agc_amd/dist.py).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        -m agc_amd.dist_create [-k 31] [-l 20] [-s 60000] [-b 50] [-a] [-c] [-t threads] -o out.agc ref.fa s1.fa s2.fa.gz ...

Options as `agc create`.  Rank r reads, uploads and classifies the files r, r+N, ... (-c: the registration units r, r+N, ... --
runs of -b contigs across the files); rank 0 writes the archive, which is
(--append in.agc: every rank loads in.agc, all files are new samples, k / l / s / b come from the archive, as `agc append`;
with -c the first unit completes the batch in.agc ended in)
byte-identical to what the single-GPU `agc_amd create` and the reference CLI write for the same command line."""
import argparse
import os
import sys


def main(argv=None):
    ap = argparse.ArgumentParser(prog="agc_amd.dist_create")
    ap.add_argument("-k", type=int, default=31)
    ap.add_argument("-l", type=int, default=20)
    ap.add_argument("-s", type=int, default=60000)
    ap.add_argument("-b", type=int, default=50)
    ap.add_argument("-a", action="store_true")
    ap.add_argument("-c", action="store_true", help="concatenated genomes: every contig is a sample")
    ap.add_argument("-t", type=int, default=0)
    ap.add_argument("-o", required=True)
    ap.add_argument("--append", default=None, metavar="IN.agc", help="append the files to this archive (as `agc append`); every rank reads it")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, one GPU per rank) or gloo (ranks may share a GPU)")
    ap.add_argument("files", nargs="+")
    a = ap.parse_args(argv)
    import numpy as np
    import torch
    import torch.distributed as dist
    from agc_amd import fasta, host
    from agc_amd.dist import DistCompressor, archive_contig_names, concatenated_units

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    if a.backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(a.backend)
    files, seen = [], set()
    for f in a.files:  # sanitize_input_file_names, application.cpp:584-601
        if f not in seen:
            seen.add(f)
            files.append(f)
    threads = a.t or max(1, (os.cpu_count() or 2) // 2)
    cmp_ = host.Compressor(local)
    cmp_.set_distributed(rank, world, 0)
    if a.append:
        cmp_.append(a.append, a.o if rank == 0 else "", concatenated=a.c, adaptive=a.a, n_threads=threads if rank == 0 else 2)
    else:
        cmp_.create(a.o if rank == 0 else "", pack_cardinality=a.b, k=a.k, ref_file=files[0], segment_size=a.s, min_match_len=a.l,
                    concatenated=a.c, adaptive=a.a, n_threads=threads if rank == 0 else 2)
    dc = DistCompressor(cmp_, dist, rank, world, device=dev)
    keep = {}

    def get_sample(i):
        keep.pop(i - world, None)  # the previous own sample has been committed by now
        names, codes, off = fasta.read_codes(files[i])
        keep[i] = torch.from_numpy(np.concatenate([codes, np.full(4096, 4, np.uint8)])).to(dev)
        torch.cuda.synchronize(dev)
        return fasta.sample_name(files[i]), names, keep[i].data_ptr(), off

    n_units = len(files)
    if a.c:
        # the reference's registration units: runs of -b contigs across the files, every contig a sample of its own (sample name "")
        # (append: the first unit completes the batch the input archive ended in; its contigs are not taken again)
        n0, names0, b0 = archive_contig_names(a.append) if a.append else (0, (), None)
        units = concatenated_units([fasta.read_codes(f)[0] for f in files], b0 or a.b, already=n0, seen=names0)
        n_units = len(units)

        def get_sample(i):  # noqa: F811
            keep.pop(i - world, None)
            names, parts, off, cache = [], [], [0], {}
            for fi, ci in units[i]:
                if fi not in cache:
                    cache = {fi: fasta.read_codes(files[fi])}
                fn, fc, fo = cache[fi]
                names.append(fn[ci])
                parts.append(fc[int(fo[ci]):int(fo[ci + 1])])
                off.append(off[-1] + parts[-1].size)
            codes = np.concatenate(parts + [np.full(4096, 4, np.uint8)])
            keep[i] = torch.from_numpy(codes).to(dev)
            torch.cuda.synchronize(dev)
            return "", names, keep[i].data_ptr(), np.asarray(off, np.uint64)

    # (-a too: a sample that needs new splitters is prepared again at its turn; append: a packed group answers Estimate with 0 until a
    # record unpacks it, so nothing can be classified ahead of its turn)
    dc.compress(n_units, get_sample, prefetch=not a.append)
    dc.close(n_threads=threads if rank == 0 else 2)  # the delta packs are entropy-coded on every rank's GPU
    cmp_.close_handle()
    dist.barrier()
    dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
