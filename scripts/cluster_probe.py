"""how well do the contigs of the previous k cluster the seeds?  python scripts/cluster_probe.py [n_reads]"""
import collections, os, subprocess, sys, tempfile
sys.path.insert(0, ".")
import numpy as np
from megagta_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
w = tempfile.mkdtemp(dir="/tmp")
mg = synth.make_metagenome(n, 150, (("rplB", 277), ("nirK", 360)), seed=1)
gl = synth.write_gene_models(mg.genes, os.path.join(w, "genes"))
synth.write_fasta(mg.reads, os.path.join(w, "reads.fa"))
out = os.path.join(w, "out")
env = dict(os.environ, MEGAGTA_CLUSTER_FILE=os.path.join(w, "clusters.txt"))
subprocess.run([sys.executable, "megagta_amd/megagta.py", "-r", os.path.join(w, "reads.fa"), "-g", gl, "-k", "30,36,45", "-o", out, "-c", "1",
                "--one-process-per-step"], check=True, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
cl = np.loadtxt(os.path.join(w, "clusters.txt"), dtype=np.int64)        # the last findstart call: nirK
seeds = open(os.path.join(out, "k44", "44_nirK_starting_kmers.txt")).read().splitlines()
print("seeds", len(seeds), "cluster lines", cl.size)
print("in a contig:", int((cl >= 0).sum()), f"({(cl >= 0).mean():.3f})")
c = collections.Counter(cl[cl >= 0].tolist())
sizes = np.array(sorted(c.values()))
print("clusters", len(c), "sizes: median", int(np.median(sizes)), "mean", round(sizes.mean(), 1), "max", int(sizes.max()),
      "seeds in clusters >= 8:", int(sizes[sizes >= 8].sum()))
print("genomes", len(mg.genes[0].variants))
