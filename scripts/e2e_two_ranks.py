#!/usr/bin/env python3
"""`megagta.py --gpus 2` at 2 M reads on ONE GPU (ranks share device 0, gloo): the rank-aware product path (buildgraph sharded into two
.sdbg files per k, graph files parsed on the device by every rank, seeds sharded by gene, one all-gather of contigs) next to `--gpus 1`
on the same files: wall seconds and whether the final contigs are the same files.  python scripts/e2e_two_ranks.py [n_reads]"""
import filecmp, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from megagta_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
d = tempfile.mkdtemp(prefix="mgta_e2e2_")
mg = synth.make_metagenome_device(n, 150, (("rplB", 277), ("nirK", 360)), seed=1000 + n % 997, device="cuda:0", host_sample=n)
gl = synth.write_gene_models(mg.genes, d + "/models")
bench.write_fasta_fast(mg.sample_reads, d + "/reads.fa")
del mg
import torch
torch.cuda.empty_cache()
res = {}
for gpus in (1, 2):
    env = dict(os.environ)
    if gpus > 1:
        env.update(MEGAGTA_DEVICE="0", MEGAGTA_DIST_BACKEND="gloo")
    t = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "megagta_amd", "megagta.py"), "-r", d + "/reads.fa", "-g", gl, "-k", "30,36,45", "-o", f"{d}/out{gpus}",
                        "-t", "16", "--gpus", str(gpus)], capture_output=True, text=True, env=env)
    res[gpus] = time.time() - t
    assert r.returncode == 0, r.stderr[-2000:]
    print(f"--gpus {gpus}: {res[gpus]:.1f} s", flush=True)
same = all(filecmp.cmp(f"{d}/out1/contigs/{g}/nucl_merged.fasta", f"{d}/out2/contigs/{g}/nucl_merged.fasta", shallow=False) for g in ("rplB", "nirK"))
raw = all(filecmp.cmp(f"{d}/out1/k44/44_raw_contigs_{g}.fasta", f"{d}/out2/k44/44_raw_contigs_{g}.fasta", shallow=False) for g in ("rplB", "nirK"))
files = sorted(f for f in os.listdir(f"{d}/out2/k44") if ".sdbg" in f)
print(f"{n} reads: filtered contigs identical: {same}; raw contigs identical: {raw} (two genes on two ranks: every gene searched by one rank); graph files of k = 44 with two ranks: {files}")
import shutil
shutil.rmtree(d, ignore_errors=True)
