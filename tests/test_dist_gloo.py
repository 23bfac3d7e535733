"""N > 1 path on CPU: world_size 2, gloo.  The shards come from the oracle (CPU) split by bucket range;
what is tested is the sharding rule and the two all-gathers (edge-stream shards, contigs)."""
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


def _worker(rank, world, port, golden_dir, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from megagta_amd import api, dist as mdist
    from oracle import oracle as O
    packed, start = readlib.load_for_build(os.path.join(golden_dir, "ragged", "reads.lib"))
    full = O.Stream.build(packed, start, 29, threads=1).edges()
    b0, b1 = mdist.bucket_share(rank, world)
    bstart = np.concatenate([[0], np.cumsum(full.bucket_items)])
    is_large = (full.records >> 8) == 255
    is_tip = ((full.records >> 5) & 1).astype(bool)
    lc, tc = np.concatenate([[0], np.cumsum(is_large)]), np.concatenate([[0], np.cumsum(is_tip)])
    r0, r1 = bstart[b0], bstart[b1]
    bi = np.zeros(65536, np.int64)
    bi[b0:b1] = full.bucket_items[b0:b1]
    local = api.EdgeStream(k=29, words_per_tip=full.words_per_tip, bucket_items=bi, records=full.records[r0:r1],
                           large=full.large[lc[r0]:lc[r1]], tips=full.tips[tc[r0] * full.words_per_tip:tc[r1] * full.words_per_tip])
    whole = mdist.all_gather_edge_stream(local)
    ok1 = whole.md5() == full.md5()
    mine = mdist.seed_share(11, rank, world)
    contigs = mdist.all_gather_contigs(11, mine, ["acgt" * (int(i) + 1) + f"x{i}" for i in mine])
    ok2 = contigs == ["acgt" * (i + 1) + f"x{i}" for i in range(11)]
    ret[rank] = (ok1, ok2, int(local.records.size))
    dist.destroy_process_group()


def test_two_rank_gather(golden_dir, oracle):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), golden_dir, ret), nprocs=world, join=True)
    assert all(ret[r][0] and ret[r][1] for r in range(world))
    assert ret[0][2] > 0 and ret[1][2] > 0          # both ranks really owned part of the stream


def test_gene_first_seed_share():
    """genes -> ranks first, then round-robin inside a gene: every seed exactly once, a rank holds one gene when there are enough ranks"""
    from megagta_amd import dist as mdist
    for world in (1, 2, 3, 4, 8):
        for per_gene in ([100, 140], [5, 5000], [7], [10, 20, 30, 40, 50], [0, 9]):
            got = [mdist.gene_seed_share(per_gene, r, world) for r in range(world)]
            for g, n in enumerate(per_gene):
                allidx = np.sort(np.concatenate([got[r][g] for r in range(world)]))
                assert allidx.tolist() == list(range(n)), (world, per_gene, g)
            if world >= len(per_gene):
                assert all(sum(1 for g in range(len(per_gene)) if got[r][g].size) <= 1 for r in range(world))
    # 8 ranks, two genes with 1 : 1.4 seeds: 3 + 5 ranks
    got = [mdist.gene_seed_share([1000, 1400], r, 8) for r in range(8)]
    assert [int(got[r][0].size > 0) for r in range(8)] == [1, 1, 1, 0, 0, 0, 0, 0]


def test_bucket_share_covers_everything():
    from megagta_amd import dist as mdist
    for world in (1, 2, 3, 4, 8):
        spans = [mdist.bucket_share(r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == 65536
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
