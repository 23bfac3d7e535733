"""A* rate on the graph of the headline size (100 M reads, 6.3 G edges) against the 10 M-read graph: python scripts/probe_astar_100m.py [n_reads] [n_seeds]"""
import os, sys, tempfile, time
sys.path.insert(0, ".")
import numpy as np
import torch  # noqa: F401
from megagta_amd import api, synth, hmm as hmmlib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
t0 = time.time()
mg = synth.make_metagenome(n, 150, (("rplB", 277),), seed=1)
packed, start = synth.pack_reads_for_build(mg.reads)
print(f"{n} reads in {time.time() - t0:.0f} s", flush=True)
ctx = api.Context(0)
stream = ctx.build_sdbg(ctx.upload_reads(packed, start), 44, collect=True)
g = api.Graph(ctx, stream)
print(f"{g.size} edges", flush=True)
td = tempfile.mkdtemp()
synth.write_gene_models(mg.genes, td)
fw = api.DeviceHmm(ctx, hmmlib.parse_hmm(os.path.join(td, "rplB", "for_enone.hmm")))
rv = api.DeviceHmm(ctx, hmmlib.parse_hmm(os.path.join(td, "rplB", "rev_enone.hmm")))
seeds = synth.synthetic_seeds(mg.genes[0], 45, ns, seed=4)
for rep in range(2):
    res, st = api.astar_search(g, fw, rv, [x[0] for x in seeds], [x[1] - 1 for x in seeds], 20, 0.5)
    e = np.array([[r.right_side["n_expanded"], r.left_side["n_expanded"]] for r in res]).reshape(-1)
    print(f"rep {rep}: {ns} seeds, {st['n_expansions']} expansions, kernel {st['ms_kernel']:.0f} ms = {st['n_expansions'] / st['ms_kernel'] / 1e3:.1f} M/s; "
          f"per search mean {e.mean():.0f} p99 {np.percentile(e, 99):.0f} max {e.max()}; retries {st['n_retries']}", flush=True)
