"""`denovo` at scale against the reference binary on the same graph files: python scripts/bench_denovo.py [n_genomes] [genome_len] [k] [min_count]
(reads = 2 strains x n_genomes x 20x coverage).  Prints our device times, the reference's wall time with 1 thread and with all cores, and
whether our contigs equal the reference's one-thread contigs byte for byte."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, ".")
from megagta_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "megagta_amd", "bin", "megagta")
REF = os.path.join(ROOT, "oracle", "_ref", "megagta")
ng = int(sys.argv[1]) if len(sys.argv) > 1 else 50
gl = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 29
mc = sys.argv[4] if len(sys.argv) > 4 else "2"
w = tempfile.mkdtemp(dir="/tmp")
t0 = time.time()
reads = synth.make_strain_mix(5, n_genomes=ng, genome_len=gl, read_len=100, cov=20, snp_every=120, err=0.003)
with open(os.path.join(w, "reads.fa"), "w") as f:
    for i, r in enumerate(reads):
        f.write(f">r{i}\n{''.join('ACGT'[c] for c in r)}\n")
with open(os.path.join(w, "reads.lib"), "w") as f:
    f.write(f"reads.fa\nse {os.path.join(w, 'reads.fa')}\n")
print(f"{len(reads)} reads written in {time.time() - t0:.1f} s", flush=True)
run = lambda cmd: subprocess.run(cmd, check=True, capture_output=True, text=True)
run([BIN, "buildlib", os.path.join(w, "reads.lib"), os.path.join(w, "reads.lib")])
r = run([BIN, "buildgraph", "-k", str(k), "-m", mc, "--host_mem", "64000000000", "--mem_flag", "1", "--num_cpu_threads", "32", "--num_output_threads", "8",
         "--read_lib_file", os.path.join(w, "reads.lib"), "--output_prefix", os.path.join(w, "g")])
print([l for l in r.stderr.splitlines() if "device build" in l][-1:], flush=True)
t0 = time.time()
r = run([BIN, "denovo", "-s", os.path.join(w, "g"), "-o", os.path.join(w, "ours"), "--max_tip_len", "150", "--min_contig", str(k + 7)])
t_ours = time.time() - t0
print("ours: wall %.2f s |" % t_ours, [l for l in r.stderr.splitlines() if "Tips removed" in l or "Number of Edges" in l], flush=True)
ncpu = os.cpu_count()
for threads in (ncpu, 1):
    t0 = time.time()
    r = subprocess.run([REF, "denovo", "-s", os.path.join(w, "g"), "-o", os.path.join(w, f"ref{threads}"), "-t", str(threads), "--max_tip_len", "150",
                        "--min_contig", str(k + 7)], capture_output=True, text=True, timeout=3000)
    print(f"reference -t {threads}: wall {time.time() - t0:.2f} s |", [l.split("]")[-1].strip() for l in r.stderr.splitlines() if "Time elapsed" in l or "time for building" in l], flush=True)
a = open(os.path.join(w, "ours.contigs.fa")).read()
b = open(os.path.join(w, "ref1.contigs.fa")).read()
c = open(os.path.join(w, f"ref{ncpu}.contigs.fa")).read()
print("ours == reference -t 1:", a == b, "| info", open(os.path.join(w, "ours.contigs.fa.info")).read().strip(), "|", open(os.path.join(w, "ref1.contigs.fa.info")).read().strip())
print(f"reference -t {ncpu} == reference -t 1 (as sets of sequences):", sorted(b.splitlines()[1::2]) == sorted(c.splitlines()[1::2]))
