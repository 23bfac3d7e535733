"""N > 1 path on CPU: world_size 2 and 8, gloo.  The shards come from the oracle (CPU) split by bucket range; what is tested is the
sharding rules (buckets, genes -> ranks -> seeds), the two exchanges (record shards, contigs) and the FASTA writer of the ranks' driver."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from megagta_amd import readlib


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_contig(gene: int, seed: int) -> bytes:
    """what a rank 'finds' for seed `seed` of gene `gene`: any function of the two, ragged lengths incl. empty"""
    n = (seed * 7 + gene * 3) % 23
    return (b"acgt" * n)[: 4 * n - (seed % 3 if n else 0)] + b"g%ds%d" % (gene, seed) if (seed + gene) % 11 else b""


def _worker2(rank, world, port, golden_dir, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from megagta_amd import dist as mdist
    from oracle import oracle as O
    packed, start = readlib.load_for_build(os.path.join(golden_dir, "ragged", "reads.lib"))
    full = O.Stream.build(packed, start, 29, threads=1).edges()
    b0, b1 = mdist.bucket_share(rank, world)
    bstart = np.concatenate([[0], np.cumsum(full.bucket_items)])
    r0, r1 = bstart[b0], bstart[b1]
    shard = torch.from_numpy(full.records[r0:r1].astype("<u2").view(np.uint8).copy())
    whole = mdist.all_gather_record_shards(shard).numpy().view("<u2")
    ok1 = np.array_equal(whole, full.records)
    mine = mdist.gene_seed_share([11], rank, world)[0]
    blobs = [_fake_contig(0, int(i)) for i in mine]
    offs = np.concatenate([[0], np.cumsum([len(b) for b in blobs])]).astype(np.int64)
    c, o = mdist.all_gather_packed_contigs(11, mine, np.frombuffer(b"".join(blobs), dtype=np.uint8), offs)
    ok2 = [x.encode() for x in mdist.contig_list(c, o)] == [_fake_contig(0, i) for i in range(11)]
    ret[rank] = (ok1, ok2, int(r1 - r0))
    dist.destroy_process_group()


def test_two_rank_gather(golden_dir, oracle):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker2, args=(world, _free_port(), golden_dir, ret), nprocs=world, join=True)
    assert all(ret[r][0] and ret[r][1] for r in range(world))
    assert ret[0][2] > 0 and ret[1][2] > 0          # both ranks really owned part of the stream


GENES10 = [("g%d" % i, n) for i, n in enumerate([0, 1, 7, 40, 40, 133, 257, 300, 999, 2500])]     # BASELINE config 5: ten genes, eight ranks


def _worker8(rank, world, port, tmp, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from megagta_amd import dist as mdist, search_dist
    share = mdist.gene_seed_share([n for _, n in GENES10], rank, world)
    took = 0
    calls = []
    real = dist.all_gather_into_tensor
    dist.all_gather_into_tensor = lambda out, inp, group=None: (calls.append(int(inp.numel())), real(out, inp, group=group))[1]
    results = []
    for gi, (name, n) in enumerate(GENES10):                          # the loop of search_dist.main with the search replaced by _fake_contig
        mine = share[gi]
        took += int(mine.size)
        blobs = [_fake_contig(gi, int(i)) for i in mine]
        offs = np.concatenate([[0], np.cumsum([len(b) for b in blobs])]).astype(np.int64)
        results.append((mine, np.frombuffer(b"".join(blobs), dtype=np.uint8), offs))
    merged = mdist.all_gather_all_genes([n for _, n in GENES10], [r[0] for r in results], [r[1] for r in results], [r[2] for r in results])
    dist.all_gather_into_tensor = real
    assert len(calls) == 2 and calls[0] == 1, calls                   # ONE all-gather of contigs for all ten genes (+ the 8-byte one that tells the sizes)
    if rank == 0:
        for (name, n), (c, o) in zip(GENES10, merged):
            search_dist.write_fasta(os.path.join(tmp, f"out_raw_contigs_{name}.fasta"), name, c, o)
    ret[rank] = (took, sum(1 for x in share if x.size))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [8, 4])
def test_ten_genes_on_eight_and_four_ranks(tmp_path, world):
    """config 5's shape on the CPU: ten genes of very different seed counts over 8 ranks (fewer ranks than genes: whole genes are dealt,
    heaviest first) and over 4; every seed is searched exactly once, ONE all-gather of contigs for the whole run (north_star), and the files rank 0 writes are the ones
    a single process would write (record names and order of hmm_graph_search.h:79)"""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker8, args=(world, _free_port(), str(tmp_path), ret), nprocs=world, join=True)
    assert sum(ret[r][0] for r in range(world)) == sum(n for _, n in GENES10)
    loads = [ret[r][0] for r in range(world)]
    assert max(loads) <= 2500 + 1 and min(loads) > 0, loads       # the heaviest gene has a rank to itself; nobody idles
    for gi, (name, n) in enumerate(GENES10):
        want = b"".join(b">%s_contig_%d_contig_%d\n%s\n" % (name.encode(), 2 * i, 2 * i + 1, _fake_contig(gi, i)) for i in range(n))
        assert (tmp_path / f"out_raw_contigs_{name}.fasta").read_bytes() == want, name


def test_gene_first_seed_share():
    """genes -> ranks first, then round-robin inside a gene: every seed exactly once, a rank holds one gene when there are enough ranks"""
    from megagta_amd import dist as mdist
    for world in (1, 2, 3, 4, 8, 16):
        for per_gene in ([100, 140], [5, 5000], [7], [10, 20, 30, 40, 50], [0, 9], [n for _, n in GENES10]):
            got = [mdist.gene_seed_share(per_gene, r, world) for r in range(world)]
            for g, n in enumerate(per_gene):
                allidx = np.sort(np.concatenate([got[r][g] for r in range(world)]))
                assert allidx.tolist() == list(range(n)), (world, per_gene, g)
            if world >= len(per_gene):
                assert all(sum(1 for g in range(len(per_gene)) if got[r][g].size) <= 1 for r in range(world))
            else:                                                     # fewer ranks than genes: a gene is never split
                assert all(sum(1 for r in range(world) if got[r][g].size) <= 1 for g in range(len(per_gene)))
    # 8 ranks, two genes with 1 : 1.4 seeds: 3 + 5 ranks
    got = [mdist.gene_seed_share([1000, 1400], r, 8) for r in range(8)]
    assert [int(got[r][0].size > 0) for r in range(8)] == [1, 1, 1, 0, 0, 0, 0, 0]
    # config 4's shape, five genes on eight ranks: every gene has its own ranks, the heavy ones more
    got = [mdist.gene_seed_share([100, 100, 100, 400, 800], r, 8) for r in range(8)]
    assert [sum(1 for r in range(8) if got[r][g].size) for g in range(5)] == [1, 1, 1, 2, 3]
    # five genes on four ranks: the two lightest share a rank
    got = [mdist.gene_seed_share([500, 400, 300, 200, 100], r, 4) for r in range(4)]
    assert sorted(sum(int(x.size) for x in got[r]) for r in range(4)) == [300, 300, 400, 500]


def test_bucket_share_covers_everything():
    from megagta_amd import dist as mdist
    for world in (1, 2, 3, 4, 8):
        spans = [mdist.bucket_share(r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == 65536
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def _worker_pieces(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from megagta_amd import dist as mdist
    rng = np.random.default_rng(100 + rank)
    sizes = [0, 70_001, 3, 1000][:world]                              # one rank with nothing, one whose payload is many pieces long
    mine = torch.from_numpy(rng.integers(0, 256, sizes[rank], dtype=np.uint8))
    calls = []
    real = dist.all_gather_into_tensor
    dist.all_gather_into_tensor = lambda out, inp, group=None: (calls.append((int(inp.numel()), int(out.numel()))), real(out, inp, group=group))[1]
    parts = mdist.all_gather_bytes(mine, piece=1000, to_host=True)
    dist.all_gather_into_tensor = real
    ok = [p.numel() for p in parts] == sizes
    for r in range(world):
        want = np.random.default_rng(100 + r).integers(0, 256, sizes[r], dtype=np.uint8)
        ok = ok and np.array_equal(parts[r].numpy(), want)
    # the sizes first, then ceil(longest / piece) collectives of `piece` bytes per rank: the longest payload sets their NUMBER, no buffer's size
    ok = ok and calls[0] == (1, world) and len(calls) == 1 + 71 and all(c == (1000, world * 1000) for c in calls[1:])
    ret[rank] = ok
    dist.destroy_process_group()


def test_variable_length_gather_moves_pieces_not_the_longest_payload():
    """advisor r4: padding every rank's payload to the longest made one rank's blob (a whole gene's contigs) set world x longest bytes on
    every rank; the exchange now moves pieces of a fixed size and every rank's bytes land in a tensor of exactly their length"""
    world = 4
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_pieces, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world))
