#!/usr/bin/env python3
"""Which pool sizes of tests/test_astar_gpu.py::test_window_mode_with_a_starved_pool_is_still_the_roomy_result make searches YIELD in place
(n_retries), resume (n_resumes), use the reserve: python scripts/probe_starved_pools.py "window:rate:kb,kb,...;..." """
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from megagta_amd import api, synth, hmm as hmmlib
spec = sys.argv[1] if len(sys.argv) > 1 else "8:0:5120,6144,7168,9216,10240;64:4:6144,8192,10240,12288;4:0:6144,8192"
mg = synth.make_metagenome(20000, 150, (("rplB", 120),), seed=9, reads_per_genome=1000)
packed, start = synth.pack_reads_for_build(mg.reads)
ctx = api.Context(0)
stream = ctx.build_sdbg(ctx.upload_reads(packed, start), 44)
with tempfile.TemporaryDirectory() as td:
    synth.write_gene_models(mg.genes, td)
    fw = api.DeviceHmm(ctx, hmmlib.parse_hmm(os.path.join(td, "rplB", "for_enone.hmm")))
    rv = api.DeviceHmm(ctx, hmmlib.parse_hmm(os.path.join(td, "rplB", "rev_enone.hmm")))
    seeds = synth.synthetic_seeds(mg.genes[0], 45, 400, seed=4)
    g = api.Graph(ctx, stream)
    kmers, states = [s[0] for s in seeds], [s[1] - 1 for s in seeds]
    for part in spec.split(";"):
        w, r, kbs = part.split(":")
        want, st0 = api.astar_search(g, fw, rv, kmers, states, 0, 0.5, cache_mode=int(w), cost_rate=int(r))
        for kb in kbs.split(","):
            ctx.set_search_arena(7, int(kb) << 10)
            try:
                got, st = api.astar_search(g, fw, rv, kmers, states, 0, 0.5, cache_mode=int(w), cost_rate=int(r))
                same = all(a.contig(k) == b.contig(k) for a, b, k in zip(got, want, kmers))
                print(f"window {w} rate {r} pool {kb} KB: yields {st['n_retries']}, resumes {st['n_resumes']}, reserve used {st['reserve_used']}, same contigs {same}, "
                      f"expansions equal {st['n_expansions'] == st0['n_expansions']}", flush=True)
            except api.MegaGtaError as e:
                print(f"window {w} rate {r} pool {kb} KB: error {str(e)[:100]}", flush=True)
            finally:
                ctx.set_search_arena(0, 0)
