#!/usr/bin/env python3
"""What one rank of an N-rank sharded build does, measured on ONE GPU: the bench's 100 M reads, then `build_sdbg` over the bucket share of
rank 0 of N = 1, 2, 4, 8 (megagta_amd.dist.bucket_share), three times each: phase times of the last one.  The all-gather of the record
shards is not in these numbers (it needs the other ranks).  python scripts/bench_shard_share.py [n_reads]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from megagta_amd import api, synth
from megagta_amd import dist as mdist

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
k = 44
mg = synth.make_metagenome_device(n, 150, (("rplB", 277), ("nirK", 360)), seed=1, device="cuda:0", host_sample=1)
ctx = api.Context(0)
rd = ctx.adopt_reads(mg.packed.data_ptr(), mg.n_words, mg.start.data_ptr(), mg.n_reads, keepalive=(mg.packed, mg.start))
ctx.keep_stream(True)
for world in (1, 2, 4, 8):
    for rank in sorted({0, world - 1}):
        b0, b1 = mdist.bucket_share(rank, world)
        for it in range(3):
            t = time.time()
            g = ctx.build_sdbg(rd, k, collect=False, bucket_range=(b0, b1))
            torch.cuda.synchronize()
            wall = time.time() - t
        s = g.stats
        print(json.dumps({"world": world, "rank": rank, "buckets": [b0, b1], "wall_ms": round(wall * 1e3, 1), "passes": s["n_passes"], "items": s["n_items"], "edges": s["n_edges"],
                          **{p: round(s[p], 1) for p in ("ms_count", "ms_gen", "ms_sort", "ms_emit", "ms_total")}}), flush=True)
