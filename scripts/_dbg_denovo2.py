import sys, os
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
os.environ["MGTA_DENOVO_DEBUG"] = "1"
api.Graph(ctx, st.edges()).denovo(150, True, 0)
