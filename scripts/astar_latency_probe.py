"""per-wave expansion latency: few concurrent searches (<< resident waves), small vs large graph"""
import sys, time, json, os, tempfile
sys.path.insert(0, '.')
import numpy as np
from megagta_amd import api, synth, hmm as hmmlib
for n, M in ((20000, 277), (1000000, 277)):
    mg = synth.make_metagenome(n, 150, (("rplB", M),), seed=1, reads_per_genome=2000 if n > 100000 else 1000)
    packed, start = synth.pack_reads_for_build(mg.reads)
    ctx = api.Context(0)
    stream = ctx.build_sdbg(ctx.upload_reads(packed, start), 44)
    g = api.Graph(ctx, stream)
    td = tempfile.mkdtemp()
    synth.write_gene_models(mg.genes, td)
    fw = api.DeviceHmm(ctx, hmmlib.parse_hmm(os.path.join(td, "rplB", "for_enone.hmm")))
    rv = api.DeviceHmm(ctx, hmmlib.parse_hmm(os.path.join(td, "rplB", "rev_enone.hmm")))
    seeds = synth.synthetic_seeds(mg.genes[0], 45, 64, seed=4)
    for ns in (8, 64):
        res, st = api.astar_search(g, fw, rv, [s[0] for s in seeds[:ns]], [s[1] - 1 for s in seeds[:ns]], 20, 0.5)
        longest = max(max(r.right_side["n_expanded"], r.left_side["n_expanded"]) for r in res)
        print(json.dumps({"reads": n, "edges": int(stream.records.size), "seeds": ns, "expansions": st["n_expansions"], "ms_kernel": round(st["ms_kernel"], 2),
                          "longest_search_expansions": longest, "us_per_expansion_of_longest(upper bound)": round(st["ms_kernel"] * 1e3 / max(1, longest), 2),
                          "retries": st["n_retries"]}), flush=True)
    ctx.close()
