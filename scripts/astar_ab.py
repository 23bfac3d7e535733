"""A/B bench of the A* kernel on the metric's graph (as bench.py's search leg sets it up: reads generated on the device, stream kept, graph
resident, the build's scratch released): cold searches of rplB + nirK, then -- with PRODUCT > 0 -- findstart's seeds under the ordered window.
MEGAGTA_HIP_LIB selects the library variant.  python scripts/astar_ab.py [n_reads] [seeds per gene] [product seeds per gene] [reps]"""
import os, sys, tempfile, time, json
sys.path.insert(0, ".")
import numpy as np
import torch  # noqa: F401
from megagta_amd import api, synth, hmm as hmmlib
from megagta_amd import findstart as fsm
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
nprod = int(sys.argv[3]) if len(sys.argv) > 3 else 0
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 1
k1 = 45
t0 = time.time()
mg = synth.make_metagenome_device(n, 150, (("rplB", 277), ("nirK", 360)), seed=1, device="cuda:0", host_sample=1_000_000 if nprod else 1)
ctx = api.Context(0)
rd = ctx.adopt_reads(mg.packed.data_ptr(), mg.n_words, mg.start.data_ptr(), mg.n_reads, keepalive=(mg.packed, mg.start))
ctx.keep_stream(True)
ctx.build_sdbg(rd, k1 - 1, collect=False)
graph = api.Graph(ctx, None, k1 - 1)
ctx.keep_stream(False)
ctx.release_scratch()
print(f"[{time.time() - t0:.0f} s] {n} reads, graph of {graph.size} edges resident; lib {os.environ.get('MEGAGTA_HIP_LIB', 'default')}", flush=True)
td = tempfile.mkdtemp()
synth.write_gene_models(mg.genes, td)
hm, seeds, prod = [], [], []
for gi, gene in enumerate(mg.genes):
    d = os.path.join(td, gene.name)
    hm.append((api.DeviceHmm(ctx, hmmlib.parse_hmm(os.path.join(d, "for_enone.hmm"))), api.DeviceHmm(ctx, hmmlib.parse_hmm(os.path.join(d, "rev_enone.hmm")))))
    seeds.append(synth.synthetic_seeds(gene, k1, ns, seed=4 + gi))
    if nprod:
        fwords, fpos = fsm.reference_words(os.path.join(d, "ref_aligned.faa"), k1 // 3)
        fhits, _ = fsm.find_hits(ctx, rd, True, k1, fsm.pack_words(fwords, k1 // 3))
        ps = bench.product_seed_list(fhits, mg.sample_reads, fwords, fpos, k1)
        lo = max(0, (len(ps) - nprod) // 2)
        prod.append(ps[lo:lo + nprod])
for rep in range(reps + 1):                          # (the first pass obtains the pool and loads the kernels: not reported as a result)
    tot_e, tot_ms = 0, 0.0
    for gi in range(len(mg.genes)):
        want = rep == 0 and os.environ.get("ASTAR_AB_LENGTHS")
        out_ = api.astar_search_packed(graph, hm[gi][0], hm[gi][1], [s[0] for s in seeds[gi]], [s[1] - 1 for s in seeds[gi]], 20, 0.5, want_sides=bool(want))
        st = out_[2]
        if want:
            # how long a search is against what is known before it runs: the model columns its two sides have to cover (the host's LPT guess)
            sides = out_[3]
            M = mg.genes[gi].M if hasattr(mg.genes[gi], "M") else (277 if gi == 0 else 360)
            e = np.array([[sides[2 * i].n_expanded, sides[2 * i + 1].n_expanded] for i in range(len(seeds[gi]))], dtype=np.float64)
            pos = np.array([s[1] - 1 for s in seeds[gi]], dtype=np.float64)
            cols = np.stack([M - pos - k1 // 3, pos], axis=1)
            for d in (0, 1):
                order = np.argsort(-cols[:, d])
                top = np.argsort(-e[:, d])[:100]
                rank_of_top = np.argsort(order).astype(np.int64)[top]
                print(f"    {mg.genes[gi].name} dir {d}: corr(columns, expansions) {np.corrcoef(cols[:, d], e[:, d])[0, 1]:.3f}; the 100 longest searches "
                      f"(max {e[:, d].max():.0f}, mean {e[:, d].mean():.0f}) sit at median place {np.median(rank_of_top):.0f} of {len(order)} in a columns-descending order "
                      f"(worst {rank_of_top.max()})", flush=True)
        if rep == 0 and os.environ.get("ASTAR_AB_LONE"):
            # the latency of ONE search: the gene's longest search run alone (nothing else on the device), microseconds per expansion
            sides = api.astar_search_packed(graph, hm[gi][0], hm[gi][1], [s[0] for s in seeds[gi]], [s[1] - 1 for s in seeds[gi]], 20, 0.5, want_sides=True)[3]
            e = np.array([max(sides[2 * i].n_expanded, sides[2 * i + 1].n_expanded) for i in range(len(seeds[gi]))])
            j = int(np.argmax(e))
            for _ in range(2):
                _, _, s1 = api.astar_search_packed(graph, hm[gi][0], hm[gi][1], [seeds[gi][j][0]], [seeds[gi][j][1] - 1], 20, 0.5)
                print(f"    LONE {mg.genes[gi].name}: seed {j}, {s1['n_expansions']} expansions (longest side {s1['max_search_expansions']}), kernel {s1['ms_kernel']:.0f} ms = "
                      f"{s1['ms_kernel'] * 1e3 / max(1, s1['max_search_expansions']):.2f} us per expansion of the longest side", flush=True)
        tot_e += st["n_expansions"]; tot_ms += st["ms_kernel"]
        print(f"  rep {rep} {mg.genes[gi].name}: {st['n_expansions']} expansions, kernel {st['ms_kernel']:.0f} ms = {st['n_expansions'] / st['ms_kernel'] / 1e3:.1f} M/s, "
              f"retries {st['n_retries']}, grown {st['n_grown']}, max search {st['max_search_expansions']}", flush=True)
    print(f"{'warm-up' if rep == 0 else 'COLD'} rep {rep}: {tot_e} expansions in {tot_ms:.0f} ms of kernel = {tot_e / tot_ms / 1e3:.1f} M expansions/s", flush=True)
if nprod:
    from megagta_amd import search_dist as sdm
    for rep in range(reps):
        tot_e, tot_ms = 0, 0.0
        for gi, gene in enumerate(mg.genes):
            ps = prod[gi]
            window, rate = sdm.window_and_rate(len(ps))
            _, _, st = api.astar_search_packed(graph, hm[gi][0], hm[gi][1], [x[0] for x in ps], [x[1] - 1 for x in ps], 20, 0.5, cache_mode=window, cost_rate=rate)
            tot_e += st["n_expansions"]; tot_ms += st["ms_total"]
            print(f"  product {gene.name}: {len(ps)} seeds, window {window} rate {rate}: {st['n_expansions']} expansions, {st['ms_total']:.0f} ms, max search {st['max_search_expansions']}", flush=True)
        print(f"PRODUCT rep {rep}: {tot_e} expansions in {tot_ms:.0f} ms = {tot_e / tot_ms / 1e3:.1f} M expansions/s", flush=True)
