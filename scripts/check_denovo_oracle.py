"""oracle denovo vs the reference binary run with one thread (build container only):
python scripts/check_denovo_oracle.py [seed] [k] [n_genomes] [min_count]"""
import os, subprocess, sys, tempfile
sys.path.insert(0, ".")
import numpy as np
from oracle import oracle

REF = os.path.join("oracle", "_ref", "megagta")


from megagta_amd.synth import make_strain_mix as strain_reads


def write_fasta(reads, path):
    with open(path, "w") as f:
        for i, r in enumerate(reads):
            f.write(f">r{i}\n{''.join('ACGT'[c] for c in r)}\n")


def reference_run(w, k, min_count, threads=1, min_contig=0, max_tip_len=150, no_bubble=False):
    run = lambda cmd: subprocess.run(cmd, check=True, capture_output=True)
    with open(os.path.join(w, "reads.lib"), "w") as f:
        f.write(f"reads.fa\nse {os.path.join(w, 'reads.fa')}\n")
    run([REF, "buildlib", os.path.join(w, "reads.lib"), os.path.join(w, "reads.lib")])
    run([REF, "buildgraph", "-k", str(k), "-m", str(min_count), "--host_mem", "4000000000", "--mem_flag", "1", "--gpu_mem", "0",
         "--num_cpu_threads", "4", "--num_output_threads", "1", "--read_lib_file", os.path.join(w, "reads.lib"), "--output_prefix", os.path.join(w, "g")])
    cmd = [REF, "denovo", "-s", os.path.join(w, "g"), "-o", os.path.join(w, "g"), "-t", str(threads), "--max_tip_len", str(max_tip_len),
           "--min_contig", str(min_contig)] + (["--no_bubble"] if no_bubble else [])
    r = subprocess.run(cmd, check=True, capture_output=True, text=True)
    return open(os.path.join(w, "g.contigs.fa")).read(), open(os.path.join(w, "g.contigs.fa.info")).read(), r.stderr


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 29
    ng = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    mc = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    tricky = len(sys.argv) > 5 and sys.argv[5] == "tricky"
    snp = int(sys.argv[6]) if len(sys.argv) > 6 else 150
    w = tempfile.mkdtemp(dir="/tmp/dn")
    write_fasta(strain_reads(seed, ng, tricky=tricky, snp_every=snp), os.path.join(w, "reads.fa"))
    ref, info, log = reference_run(w, k, mc, min_contig=k + 2)
    g = oracle.Graph(oracle.Stream.read(os.path.join(w, "g")))
    mine, st = g.denovo(150, False, k + 2)
    print([l for l in log.splitlines() if "tips removed" in l][-1:], [l for l in log.splitlines() if "bubbles" in l])
    print("oracle", st, "info", info.strip(), "equal", mine == ref)
    if mine != ref:
        a, b = ref.splitlines(), mine.splitlines()
        print(len(a), len(b), "same sorted seqs:", sorted(a[1::2]) == sorted(b[1::2]))
        for i, (x, y) in enumerate(zip(a, b)):
            if x != y:
                print(i, x[:100], "|", y[:100])
                break
