"""reads -> graph -> seeds -> contigs in one process on one GPU (build, device-resident graph, findstart, windowed search):
python scripts/bench_pipeline_10m.py [n_reads] [window]"""
import json, os, sys, tempfile, time
sys.path.insert(0, ".")
import torch  # noqa: F401
from megagta_amd import api, findstart, synth, hmm as hmmlib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
def _cfg(x):   # window[:lanes per search[:cost rate]]
    a = [int(y) for y in x.split(":")]
    return (a + [16, 0][len(a) - 1:])[:3]
configs = [_cfg(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "4096").split(",")]
mg = synth.make_metagenome(n, 150, (("rplB", 277),), seed=1)
td = tempfile.mkdtemp()
synth.write_gene_models(mg.genes, td)
faa = os.path.join(td, "rplB", "ref_aligned.faa")
packed, start = synth.pack_reads_for_build(mg.reads)
ctx = api.Context(0)
rd = ctx.upload_reads(packed, start)
ctx.build_sdbg(rd, 44, collect=False)          # warm the pool
t0 = time.time()
s = ctx.build_sdbg(rd, 44, collect=False).stats
g = api.Graph(ctx, None, 44)
t1 = time.time()
words, mpos = findstart.reference_words(faa, 15)
hits, ms_fs = findstart.find_hits(ctx, rd, True, 45, findstart.pack_words(words, 15))
t2 = time.time()
lines = findstart.seed_lines(hits, lambda i: "".join("ACGT"[x] for x in mg.reads[i]), words, mpos, 45)
t3 = time.time()
seeds = [(l.split("\t")[3], int(l.split("\t")[7])) for l in lines]
fw = api.DeviceHmm(ctx, hmmlib.parse_hmm(os.path.join(td, "rplB", "for_enone.hmm")))
rv = api.DeviceHmm(ctx, hmmlib.parse_hmm(os.path.join(td, "rplB", "rev_enone.hmm")))
for window, group, rate in configs:
  os.environ["MGTA_ASTAR_GROUP"] = str(group)
  t3 = time.time()
  res, st = api.astar_search(g, fw, rv, [x[0] for x in seeds], [x[1] - 1 for x in seeds], 20, 0.5, cache_mode=window, cost_rate=rate)
  t4 = time.time()
  print(json.dumps({"reads": n, "build_ms": round(s["ms_total"], 1), "build+graph_s": round(t1 - t0, 3), "findstart_kernel_ms": round(ms_fs, 1), "hits": int(hits.size),
                  "seeds": len(seeds), "seed_lines_host_s": round(t3 - t2, 2), "search_kernel_s": round(st["ms_kernel"] / 1e3, 2), "search_wall_s": round(t4 - t3, 2),
                  "expansions": st["n_expansions"], "window": window, "group": group, "cost_rate": rate, "grown": st["n_grown"], "pool_used_GB": round(st["pool_used"] / 1e9, 1),
                  "distinct_contigs": len({r.contig(x[0]) for r, x in zip(res, seeds)})}), flush=True)
