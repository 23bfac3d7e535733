"""Multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

The path shards in two places:
  * SdBG build: the 65536 prefix buckets are independent once every rank holds the reads -> rank r
    builds buckets [r*S, (r+1)*S); ONE all-gather of the record shards makes the graph whole everywhere;
  * A* search: seeds are independent given the (replicated) graph -> seeds are dealt round-robin,
    ONE all-gather of the contig bytes at the end; rank 0 writes FASTA in seed order.
No collective runs inside any kernel.  Works with any backend (gloo on CPU tensors in the tests).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from .api import NUM_BUCKETS, EdgeStream


def bucket_share(rank: int, world: int) -> tuple[int, int]:
    s = (NUM_BUCKETS + world - 1) // world
    return min(NUM_BUCKETS, rank * s), min(NUM_BUCKETS, (rank + 1) * s)


def _dev():
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def _all_gather_var(arr: np.ndarray, group=None) -> list[np.ndarray]:
    """all-gather of variable-length 1-D arrays: lengths first, then payloads (as bytes) padded to the maximum"""
    world = dist.get_world_size(group)
    dev = _dev()
    flat = torch.from_numpy(np.ascontiguousarray(arr).view(np.uint8).copy()).to(dev)
    n = torch.tensor([flat.numel()], dtype=torch.int64, device=dev)
    ns = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(ns, n, group=group)
    sizes = [int(x.item()) for x in ns]
    mx = max(max(sizes), 1)
    pad = torch.zeros(mx, dtype=flat.dtype, device=dev)
    pad[: flat.numel()] = flat
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return [o[:s].cpu().numpy().view(arr.dtype).copy() for o, s in zip(out, sizes)]


def all_gather_edge_stream(local: EdgeStream, group=None) -> EdgeStream:
    """Every rank passes the stream of ITS bucket range (other buckets empty); every rank gets the whole stream."""
    world = dist.get_world_size(group)
    recs = _all_gather_var(local.records, group)
    large = _all_gather_var(local.large, group)
    tips = _all_gather_var(local.tips, group)
    counts = np.stack([local.bucket_items, local.bucket_large if local.bucket_large is not None else np.zeros(NUM_BUCKETS, np.int64),
                       local.bucket_tips if local.bucket_tips is not None else np.zeros(NUM_BUCKETS, np.int64)]).astype(np.int64)
    t = torch.from_numpy(counts).to(_dev())
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)          # shards are disjoint in bucket space
    counts = t.cpu().numpy()
    return EdgeStream(k=local.k, words_per_tip=local.words_per_tip, bucket_items=counts[0], records=np.concatenate(recs),
                      large=np.concatenate(large), tips=np.concatenate(tips), bucket_large=counts[1], bucket_tips=counts[2])


def seed_share(n_seeds: int, rank: int, world: int) -> np.ndarray:
    return np.arange(rank, n_seeds, world, dtype=np.int64)


def gene_seed_share(seeds_per_gene: list[int], rank: int, world: int) -> list[np.ndarray]:
    """Seeds shard by GENE first, then round-robin inside a gene (BASELINE.json north_star): with at least as many
    ranks as genes every gene gets a group of ranks (sizes proportional to its seeds, at least one each) and its seeds are dealt
    round-robin inside the group, so a rank stages ONE gene's HMM tables; with fewer ranks than genes whole genes are dealt round-robin
    over the ranks.  Returns, per gene, the seed indices this rank runs (ascending; empty for genes it does not take)."""
    n_genes = len(seeds_per_gene)
    out = [np.zeros(0, dtype=np.int64) for _ in range(n_genes)]
    if n_genes == 0:
        return out
    if world < n_genes:
        for g in range(rank, n_genes, world):
            out[g] = np.arange(seeds_per_gene[g], dtype=np.int64)
        return out
    total = max(1, sum(seeds_per_gene))
    size = [1] * n_genes                                             # ranks per gene: one each, the rest by share of the seeds
    for _ in range(world - n_genes):
        g = max(range(n_genes), key=lambda x: seeds_per_gene[x] / total / size[x])
        size[g] += 1
    first = 0
    for g in range(n_genes):
        if first <= rank < first + size[g]:
            out[g] = np.arange(rank - first, seeds_per_gene[g], size[g], dtype=np.int64)
        first += size[g]
    return out


def all_gather_contigs(n_seeds: int, mine: np.ndarray, contigs: list[str], group=None) -> list[str]:
    """mine[i] = global seed index of contigs[i]; returns all contigs in seed order on every rank.  The path's one exchange: ONE
    all-gather of a length-prefixed byte buffer per rank ([n][seed index, length]*n[bytes]), padded to the longest (a second, 8-byte
    all-gather tells the sizes)."""
    mine = np.asarray(mine, dtype=np.int64)
    enc = [c.encode() for c in contigs]
    head = np.empty(1 + 2 * mine.size, dtype=np.int64)
    head[0] = mine.size
    head[1::2] = mine
    head[2::2] = [len(b) for b in enc]
    blob = np.concatenate([head.view(np.uint8), np.frombuffer(b"".join(enc), dtype=np.uint8)])
    out = [""] * n_seeds
    for b in _all_gather_var(blob, group):
        n = int(b[:8].view(np.int64)[0])
        h = b[8:8 + 16 * n].view(np.int64)
        pos = 8 + 16 * n
        for j in range(n):
            ln = int(h[2 * j + 1])
            out[int(h[2 * j])] = b[pos:pos + ln].tobytes().decode()
            pos += ln
    return out
