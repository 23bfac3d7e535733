"""distribution of search lengths under a cache window: python scripts/astar_length_hist.py [n_reads] [window]"""
import os, sys, tempfile
sys.path.insert(0, ".")
import numpy as np
import torch  # noqa: F401
from megagta_amd import api, findstart, synth, hmm as hmmlib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
window = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
mg = synth.make_metagenome(n, 150, (("rplB", 277),), seed=1)
td = tempfile.mkdtemp()
synth.write_gene_models(mg.genes, td)
lines, _ = findstart.find_start(api.Context(0), os.path.join(td, "rplB", "ref_aligned.faa"), list(mg.reads), 45)
seeds = [(l.split("\t")[3], int(l.split("\t")[7])) for l in lines]
packed, start = synth.pack_reads_for_build(mg.reads)
ctx = api.Context(0)
ctx.build_sdbg(ctx.upload_reads(packed, start), 44, collect=False)
g = api.Graph(ctx, None, 44)
fw = api.DeviceHmm(ctx, hmmlib.parse_hmm(os.path.join(td, "rplB", "for_enone.hmm")))
rv = api.DeviceHmm(ctx, hmmlib.parse_hmm(os.path.join(td, "rplB", "rev_enone.hmm")))
for w in (0, window):
    res, st = api.astar_search(g, fw, rv, [x[0] for x in seeds], [x[1] - 1 for x in seeds], 20, 0.5, cache_mode=w)
    e = np.array([[r.right_side["n_expanded"], r.left_side["n_expanded"]] for r in res]).reshape(-1)
    print(f"window {w}: {len(seeds)} seeds, {e.sum()} expansions, kernel {st['ms_kernel']:.0f} ms, mean {e.mean():.0f}, median {np.median(e):.0f}, p90 {np.percentile(e, 90):.0f}, "
          f"p99 {np.percentile(e, 99):.0f}, max {e.max()}")
    for t in (500, 1000, 2000, 4000, 8000, 16000):
        print(f"   > {t:5d}: {100 * (e > t).mean():5.1f} % of searches, {100 * e[e > t].sum() / e.sum():5.1f} % of expansions")
