"""Golden vectors for `megagta denovo` (row f-1), made with the reference binary run with ONE thread (its multi-thread runs race):
    python tests/golden/make_golden_denovo.py          (build container only: needs oracle/_ref/megagta)
Writes tests/golden/denovo/<case>.fa.gz (the reads) and tests/golden/denovo/expected.json:
{case: {"k", "min_count", "runs": [{"max_tip_len", "no_bubble", "min_contig", "contigs": text of PREFIX.contigs.fa, "info": text of .info}]}}"""
import gzip, json, os, subprocess, sys, tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from megagta_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "megagta")
CASES = {   # name: (generator arguments, k, min_count)
    "strains_k29": (dict(seed=41, n_genomes=2, genome_len=1500), 29, 2),
    "errors_k31": (dict(seed=42, n_genomes=1, genome_len=1200, cov=12), 31, 1),
    "tricky_k21": (dict(seed=43, n_genomes=2, genome_len=1500, snp_every=40, tricky=True), 21, 2),
    "tricky_k44": (dict(seed=44, n_genomes=2, genome_len=1800, snp_every=60, tricky=True, read_len=120), 44, 2),
}
RUNS = [(150, False, None), (-1, False, 0), (150, True, 0), (0, False, 0)]   # (max_tip_len, no_bubble, min_contig; None = k + 2)


def main():
    run = lambda cmd: subprocess.run(cmd, check=True, capture_output=True)
    out = {}
    for name, (gen, k, mc) in CASES.items():
        reads = synth.make_strain_mix(**gen)
        w = tempfile.mkdtemp()
        fa = os.path.join(w, "reads.fa")
        with open(fa, "w") as f:
            for i, r in enumerate(reads):
                f.write(f">r{i}\n{''.join('ACGT'[c] for c in r)}\n")
        with open(fa, "rb") as f, gzip.GzipFile(os.path.join(HERE, "denovo", name + ".fa.gz"), "wb", mtime=0) as z:
            z.write(f.read())
        with open(os.path.join(w, "reads.lib"), "w") as f:
            f.write(f"reads.fa\nse {fa}\n")
        run([REF, "buildlib", os.path.join(w, "reads.lib"), os.path.join(w, "reads.lib")])
        run([REF, "buildgraph", "-k", str(k), "-m", str(mc), "--host_mem", "4000000000", "--mem_flag", "1", "--gpu_mem", "0", "--num_cpu_threads", "4",
             "--num_output_threads", "1", "--read_lib_file", os.path.join(w, "reads.lib"), "--output_prefix", os.path.join(w, "g")])
        runs = []
        for tip, nob, minc in RUNS:
            minc = k + 2 if minc is None else minc
            run([REF, "denovo", "-s", os.path.join(w, "g"), "-o", os.path.join(w, "o"), "-t", "1", "--max_tip_len", str(tip), "--min_contig", str(minc)]
                + (["--no_bubble"] if nob else []))
            runs.append(dict(max_tip_len=tip, no_bubble=nob, min_contig=minc, contigs=open(os.path.join(w, "o.contigs.fa")).read(),
                             info=open(os.path.join(w, "o.contigs.fa.info")).read()))
            print(name, tip, nob, minc, runs[-1]["info"].strip())
        out[name] = dict(k=k, min_count=mc, runs=runs)
    with open(os.path.join(HERE, "denovo", "expected.json"), "w") as f:
        json.dump(out, f, indent=0)


if __name__ == "__main__":
    main()
