"""Multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

The path shards in two places:
  * SdBG build: the 65536 prefix buckets are independent once every rank holds the reads -> rank r builds buckets [r*S, (r+1)*S).
    Product path (`megagta.py --gpus N`): every rank writes its share as <prefix>.sdbg.<r>, no exchange at all.  bench.py's timed step
    keeps the whole stream on every GPU instead: ONE device-to-device all-gather of the record shards (`all_gather_record_shards`), moved in
    pieces of GATHER_PIECE bytes per rank (a variable-length all-gather: the sizes differ, the collective moves equal pieces);
  * A* search: seeds are independent given the (replicated) graph -> seeds shard by GENE first, then round-robin inside a gene
    (`gene_seed_share`); ONE all-gather of the contig bytes at the end (`all_gather_packed_contigs`); rank 0 writes FASTA in seed order.
No collective runs inside any kernel.  Works with any backend (gloo on CPU tensors in the tests).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from .api import NUM_BUCKETS


def bucket_share(rank: int, world: int) -> tuple[int, int]:
    s = (NUM_BUCKETS + world - 1) // world
    return min(NUM_BUCKETS, rank * s), min(NUM_BUCKETS, (rank + 1) * s)


def _dev():
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


GATHER_PIECE = 64 << 20      # bytes per rank and collective of a variable-length exchange (staging: (world + 1) pieces)


def all_gather_bytes(flat: torch.Tensor, group=None, piece: int | None = None, to_host: bool = False) -> list[torch.Tensor]:
    """all-gather of uint8 tensors of DIFFERENT lengths.  An 8-byte all-gather tells the sizes; the payloads then travel in pieces of at
    most `piece` bytes per rank: one `all_gather_into_tensor` per piece into a staging buffer of world x piece bytes, from which every
    rank's valid bytes are copied into an output tensor of exactly that rank's size.  The longest payload decides the NUMBER of
    collectives, never the size of a buffer: with genes dealt to ranks whole, one rank's blob (the heaviest gene's contigs) used to set
    world x longest bytes of device memory on every rank (advisor r4).  `flat` may live on the host while the backend is RCCL: it is staged
    through the device piece by piece; `to_host` returns CPU tensors (the device then holds the staging only)."""
    world = dist.get_world_size(group)
    dev = _dev()
    piece = int(piece or GATHER_PIECE)
    n = torch.tensor([flat.numel()], dtype=torch.int64, device=dev)
    ns = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(ns, n, group=group)
    sizes = [int(x) for x in ns.cpu().tolist()]
    longest = max(sizes)
    out_dev = torch.device("cpu") if to_host else dev
    outs = [torch.empty(s, dtype=torch.uint8, device=out_dev) for s in sizes]
    if longest == 0:
        return outs
    plen = min(piece, longest)
    send = torch.zeros(plen, dtype=torch.uint8, device=dev)
    recv = torch.empty(world * plen, dtype=torch.uint8, device=dev)
    mine = flat.numel()
    for off in range(0, longest, plen):
        m = max(0, min(plen, mine - off))
        if m:
            send[:m].copy_(flat[off:off + m], non_blocking=False)
        dist.all_gather_into_tensor(recv, send, group=group)
        for r, s in enumerate(sizes):
            v = max(0, min(plen, s - off))
            if v:
                outs[r][off:off + v].copy_(recv[r * plen: r * plen + v])
    return outs


def all_gather_record_shards(shard: torch.Tensor, group=None) -> torch.Tensor:
    """Every rank passes the records of ITS bucket range (uint8 view of the uint16 records, on its device); every rank gets the whole
    stream in bucket order (ranks own ascending bucket ranges).  The shard never visits the host."""
    return torch.cat(all_gather_bytes(shard, group))


def gene_seed_share(seeds_per_gene: list[int], rank: int, world: int) -> list[np.ndarray]:
    """Seeds shard by GENE first, then round-robin inside a gene (BASELINE.json north_star): with at least as many
    ranks as genes every gene gets a group of ranks (sizes proportional to its seeds, at least one each) and its seeds are dealt
    round-robin inside the group, so a rank stages ONE gene's HMM tables; with fewer ranks than genes whole genes are dealt to the
    ranks, heaviest first onto the rank with the fewest seeds so far (5 genes on 4 GPUs: the two lightest share a rank).
    Returns, per gene, the seed indices this rank runs (ascending; empty for genes it does not take)."""
    n_genes = len(seeds_per_gene)
    out = [np.zeros(0, dtype=np.int64) for _ in range(n_genes)]
    if n_genes == 0:
        return out
    if world < n_genes:
        load = [0] * world
        for g in sorted(range(n_genes), key=lambda x: (-seeds_per_gene[x], x)):
            r = min(range(world), key=lambda x: (load[x], x))
            load[r] += seeds_per_gene[g]
            if r == rank:
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


def _contig_blob(mine: np.ndarray, contigs: np.ndarray, offsets: np.ndarray) -> np.ndarray:
    """one rank's contigs of one gene as bytes: [n][seed index, length] * n [contig bytes]"""
    mine = np.asarray(mine, dtype=np.int64)
    offsets = np.asarray(offsets, dtype=np.int64)
    head = np.empty(1 + 2 * mine.size, dtype=np.int64)
    head[0] = mine.size
    head[1::2] = mine
    head[2::2] = np.diff(offsets)
    return np.concatenate([head.view(np.uint8), np.ascontiguousarray(contigs, dtype=np.uint8)[: int(offsets[-1]) if offsets.size else 0]])


def _merge_contig_blobs(n_seeds: int, parts: list[np.ndarray]) -> tuple[np.ndarray, np.ndarray]:
    """the ranks' blobs of one gene -> (contigs, offsets) of all its seeds in seed order.  No Python work per seed: numpy gathers over
    pieces of at most 64 MB."""
    lens = np.zeros(n_seeds, dtype=np.int64)
    heads = []
    for b in parts:
        n = int(b[:8].view(np.int64)[0]) if b.size >= 8 else 0
        h = b[8:8 + 16 * n].view(np.int64)
        heads.append((n, h[0::2], h[1::2]))
        lens[h[0::2]] = h[1::2]
    out_off = np.zeros(n_seeds + 1, dtype=np.int64)
    np.cumsum(lens, out=out_off[1:])
    out = np.empty(int(out_off[-1]), dtype=np.uint8)
    for b, (n, idx, ln) in zip(parts, heads):
        if n == 0:
            continue
        src = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(ln, out=src[1:])
        src += 8 + 16 * n
        i = 0
        while i < n:                                                 # pieces of <= 64 MB of contig bytes
            j = int(np.searchsorted(src, src[i] + (64 << 20), side="right")) - 1
            j = min(n, max(j, i + 1))
            shift = np.repeat(out_off[idx[i:j]] - src[i:j], ln[i:j])
            pos = np.arange(src[i], src[j], dtype=np.int64)
            out[pos + shift] = b[pos]
            i = j
    return out, out_off


def all_gather_packed_contigs(n_seeds: int, mine: np.ndarray, contigs: np.ndarray, offsets: np.ndarray, group=None) -> tuple[np.ndarray, np.ndarray]:
    """mine[i] = global seed index of this rank's contig i = contigs[offsets[i]:offsets[i+1]] (uint8).  Returns (contigs, offsets) of
    ALL seeds in seed order on every rank: one gene's exchange (all_gather_all_genes is the whole run's)."""
    return all_gather_all_genes([n_seeds], [mine], [contigs], [offsets], group)[0]


def all_gather_all_genes(n_seeds: list[int], mine: list[np.ndarray], contigs: list[np.ndarray], offsets: list[np.ndarray],
                         group=None) -> list[tuple[np.ndarray, np.ndarray]]:
    """The path's ONE exchange (BASELINE.json north_star: "a single RCCL all-gather of contigs over xGMI at the end"): every rank passes, for
    every gene of the run, the contigs of the seeds it searched (mine[g][i] = seed index of contigs[g][offsets[g][i]:offsets[g][i+1]]);
    every rank gets, per gene, (contigs, offsets) of all seeds in seed order.  One buffer per rank -- [per gene: blob length][blobs] --
    and ONE all-gather of it (an 8-byte one tells the sizes)."""
    blobs = [_contig_blob(m, c, o) for m, c, o in zip(mine, contigs, offsets)]
    table = np.array([b.size for b in blobs], dtype=np.int64)
    buf = np.concatenate([table.view(np.uint8)] + blobs) if blobs else np.zeros(0, np.uint8)
    # (the rank's buffer stays on the host and so do the results: the device holds (world + 1) pieces of staging, whatever the genes weigh)
    parts = [p.numpy() for p in all_gather_bytes(torch.from_numpy(buf), group, to_host=True)]
    G = len(blobs)
    out = []
    starts = []
    for p in parts:
        t = p[:8 * G].view(np.int64)
        st = np.zeros(G + 1, dtype=np.int64)
        np.cumsum(t, out=st[1:])
        starts.append(st + 8 * G)
    for g in range(G):
        out.append(_merge_contig_blobs(n_seeds[g], [p[int(st[g]):int(st[g + 1])] for p, st in zip(parts, starts)]))
    return out


def contig_list(contigs: np.ndarray, offsets: np.ndarray) -> list[str]:
    """(tests, small batches) the packed form as Python strings"""
    raw = contigs.tobytes()
    return [raw[int(a):int(b)].decode() for a, b in zip(offsets[:-1], offsets[1:])]
