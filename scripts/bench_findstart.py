"""seed finder throughput: python scripts/bench_findstart.py [n_reads] [k] [cpu_sample_reads]
device scan (mgta_findstart kernel time) vs the reference binary's findstart (oracle/_ref, all host cores) on a sample"""
import os
import shutil
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, ".")
import torch  # noqa: F401
from megagta_amd import api, findstart, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 45
cpu_n = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
mg = synth.make_metagenome(n, 150, (("rplB", 277),), seed=1)
td = tempfile.mkdtemp(prefix="mgta_fs_")
synth.write_gene_models(mg.genes, td)
faa = os.path.join(td, "rplB", "ref_aligned.faa")
words, mpos = findstart.reference_words(faa, k // 3)
pw = findstart.pack_words(words, k // 3)
packed, start = synth.pack_reads_for_build(mg.reads)
ctx = api.Context(0)
rd = ctx.upload_reads(packed, start)
for it in range(3):
    hits, ms = findstart.find_hits(ctx, rd, True, k, pw)
    windows = n * (150 - k + 1) * 2
    print(f"device: {ms:.3f} ms, {hits.size} hits, {windows / ms / 1e6:.1f} G windows/s (both strands), {n * 150 / ms / 1e6:.1f} Gbase/s, ref words {len(words)}", flush=True)
ref = "oracle/_ref/megagta"
if os.path.exists(ref) and cpu_n > 0:
    m = min(cpu_n, n)
    synth.write_lib_bin(mg.reads[:m], os.path.join(td, "reads.lib"))
    t = time.time()
    out = subprocess.run([ref, "findstart", faa, os.path.join(td, "reads.lib.bin"), str(k), str(os.cpu_count())], capture_output=True)
    dt = time.time() - t
    print(f"reference findstart: {m} reads in {dt:.2f} s on {os.cpu_count()} threads = {m * (150 - k + 1) * 2 / dt / 1e9:.4f} G windows/s; {len(out.stdout.splitlines())} seeds", flush=True)
shutil.rmtree(td, ignore_errors=True)
