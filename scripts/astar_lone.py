"""The latency of ONE search: the longest search of a cold batch run alone on the device, microseconds per expansion (with the `prof` build of
the library, MEGAGTA_HIP_LIB=.../libmegagta_hip_prof8.so, also by phase).  python scripts/astar_lone.py [n_reads] [seeds] [gene:M]"""
import os, sys, tempfile, time
sys.path.insert(0, ".")
import numpy as np
import torch  # noqa: F401
from megagta_amd import api, synth, hmm as hmmlib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
gname, gm = (sys.argv[3].split(":") + ["360"])[:2] if len(sys.argv) > 3 else ("nirK", "360")
k1 = 45
mg = synth.make_metagenome_device(n, 150, ((gname, int(gm)),), seed=1, device="cuda:0", host_sample=1)
ctx = api.Context(0)
rd = ctx.adopt_reads(mg.packed.data_ptr(), mg.n_words, mg.start.data_ptr(), mg.n_reads, keepalive=(mg.packed, mg.start))
ctx.keep_stream(True)
ctx.build_sdbg(rd, k1 - 1, collect=False)
graph = api.Graph(ctx, None, k1 - 1)
ctx.keep_stream(False)
ctx.release_scratch()
td = tempfile.mkdtemp()
synth.write_gene_models(mg.genes, td)
d = os.path.join(td, gname)
fw, rv = api.DeviceHmm(ctx, hmmlib.parse_hmm(os.path.join(d, "for_enone.hmm"))), api.DeviceHmm(ctx, hmmlib.parse_hmm(os.path.join(d, "rev_enone.hmm")))
seeds = synth.synthetic_seeds(mg.genes[0], k1, ns, seed=4)
print(f"graph of {graph.size} edges; {ns} seeds of {gname}; lib {os.environ.get('MEGAGTA_HIP_LIB', 'default')}", flush=True)
sys.stderr.flush()
devnull = os.open(os.devnull, os.O_WRONLY)
saved = os.dup(2)
os.dup2(devnull, 2)                                   # (the batch's own profile lines are not the ones wanted)
sides = api.astar_search_packed(graph, fw, rv, [s[0] for s in seeds], [s[1] - 1 for s in seeds], 20, 0.5, want_sides=True)[3]
os.dup2(saved, 2)
e = np.array([max(sides[2 * i].n_expanded, sides[2 * i + 1].n_expanded) for i in range(ns)])
j = int(np.argmax(e))
for g in (os.environ.get("ASTAR_LONE_GROUPS", "8").split(",")):
    os.environ["MGTA_ASTAR_GROUP"] = g
    for _ in range(2):
        _, _, s1 = api.astar_search_packed(graph, fw, rv, [seeds[j][0]], [seeds[j][1] - 1], 20, 0.5)
        print(f"LONE {gname} lanes {g}: seed {j}, longest side {s1['max_search_expansions']} expansions, kernel {s1['ms_kernel']:.0f} ms = "
              f"{s1['ms_kernel'] * 1e3 / max(1, s1['max_search_expansions']):.2f} us per expansion", flush=True)
