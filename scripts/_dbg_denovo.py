import sys
sys.path.insert(0, ".")
import numpy as np, torch
from megagta_amd import api, synth, readlib
from oracle import oracle
seed = int(sys.argv[1])
rng = np.random.default_rng(500 + seed)
k = int(rng.choice([15, 21, 29, 31, 32, 44, 63, 64, 95])); mc = int(rng.choice([1, 2]))
reads = synth.make_strain_mix(900 + seed, n_genomes=int(rng.integers(2, 6)), genome_len=int(rng.integers(1500, 4000)), read_len=int(max(100, k + 40)),
                              snp_every=int(rng.choice([20, 35, 60, 150])), tricky=bool(seed & 1))
packed, start = readlib.pack_for_build(reads)
st = oracle.Stream.build(packed, start, k, threads=4) if mc == 1 else oracle.Stream.build_solid(packed, start, k, mc, False, threads=4)
ctx = api.Context(0)
print("k", k, "mc", mc)
for opts in ((150, False, k + 2), (-1, False, 0), (150, True, 0), (0, False, 0), (0, True, 0), (7, False, k + 10)):
    og = oracle.Graph(st)
    want, wst = og.denovo(*opts)
    got, gst = api.Graph(ctx, st.edges()).denovo(*opts)
    a, b = want.splitlines(), got.splitlines()
    sa, sb = sorted(a[1::2]), sorted(b[1::2])
    print(opts, got == want, wst, {x: gst[x] for x in ("n_tips", "n_bubbles", "n_contigs", "total_len", "n_paths")}, "seqs equal", sa == sb,
          "only oracle", len(set(sa) - set(sb)), "only device", len(set(sb) - set(sa)))
    if sa != sb:
        for x in list(set(sa) - set(sb))[:3]: print("  O", len(x), x[:80])
        for x in list(set(sb) - set(sa))[:3]: print("  D", len(x), x[:80])
L = oracle.lib()
for tip in (2, 3, 5, 9, 17, 150):
    og = oracle.Graph(st)
    L.orc_denovo_remove_tips(og.h, tip)
    g = api.Graph(ctx, st.edges())
    g.denovo(tip, True, 0)
    a, b = og.invalid_now(), g.invalid_bits()
    d = a ^ b
    nz = np.nonzero(d)[0]
    print("tip len", tip, "words differing", nz.size, [(int(w), hex(int(a[w])), hex(int(b[w]))) for w in nz[:4]])
