import sys, time, json, os, tempfile
sys.path.insert(0, '.')
import numpy as np
from megagta_amd import api, synth, hmm as hmmlib
n = int(sys.argv[1]); nseeds = int(sys.argv[2]); M = int(sys.argv[3]) if len(sys.argv) > 3 else 277
mg = synth.make_metagenome(n, 150, (("rplB", M),), seed=1)
packed, start = synth.pack_reads_for_build(mg.reads)
ctx = api.Context(0)
t = time.time()
stream = ctx.build_sdbg(ctx.upload_reads(packed, start), 44)
print("build+d2h s", time.time() - t, "edges", stream.records.size, flush=True)
t = time.time()
g = api.Graph(ctx, stream)
print("graph load s", time.time() - t, flush=True)
td = tempfile.mkdtemp()
synth.write_gene_models(mg.genes, td)
fw = api.DeviceHmm(ctx, hmmlib.parse_hmm(os.path.join(td, "rplB", "for_enone.hmm")))
rv = api.DeviceHmm(ctx, hmmlib.parse_hmm(os.path.join(td, "rplB", "rev_enone.hmm")))
seeds = synth.synthetic_seeds(mg.genes[0], 45, nseeds, seed=4)
for it in range(3):
    t = time.time()
    res, st = api.astar_search(g, fw, rv, [s[0] for s in seeds], [s[1] - 1 for s in seeds], 20, 0.5)
    dt = time.time() - t
    print(json.dumps(st), "wall", round(dt, 3), "Mexp/s(kernel)", st["n_expansions"] / st["ms_kernel"] / 1e3, flush=True)
lens = [len(r.left) + 45 + len(r.right) for r in res]
print("contig len mean", np.mean(lens), "max", max(lens), "full", sum(1 for r in res if r.right_side["state_no"] >= M and r.left_side["state_no"] >= M))
