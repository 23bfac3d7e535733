"""Shared helpers for the parity tests (golden-file parsing)."""
import gzip
import json
import os

import numpy as np


def gz_lines(path):
    with gzip.open(path, "rt") as f:
        return f.read().splitlines()


def load_streams(path):
    return json.load(open(path))


def fnv1a(arr: np.ndarray) -> str:
    h = 1469598103934665603
    for b in arr.tobytes():
        h ^= b
        h = (h * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h


def parse_probe_graph(lines):
    hdr = {l.split()[0]: l.split()[1:] for l in lines if not l.startswith("q ")}
    qs = []
    for l in lines:
        if not l.startswith("q "):
            continue
        t = l.split()
        q = dict(e=int(t[1]), w=int(t[3]), last=int(t[5]), tip=int(t[7]), valid=int(t[9]), multi1=int(t[11]), od=int(t[13]))
        od = max(q["od"], 0)
        q["out"] = [int(x) for x in t[14:14 + od]]
        p = 14 + od
        assert t[p] == "rl"
        q["rank_last"] = int(t[p + 1])
        q["rank_w"] = [int(x) for x in t[p + 2:p + 11]]
        assert t[p + 11] == "sl"
        q["select_last"] = int(t[p + 12])
        if len(t) > p + 13:
            assert t[p + 13] == "fwd"
            q["fwd"] = int(t[p + 14])
            q["label"] = t[p + 16]
            idn = int(t[p + 18])
            q["id"] = idn
            q["in"] = [int(x) for x in t[p + 19:p + 19 + idn]]
        qs.append(q)
    return hdr, qs


def parse_probe_hmm(lines):
    out = dict(msc={}, isc={}, tsc={}, maxm={}, h={})
    for l in lines:
        t = l.split()
        if t[0] == "M":
            out["M"] = int(t[1])
        elif t[0] == "A":
            out["A"] = int(t[1])
        elif t[0] == "alpha":
            out["alpha"] = [int(x) for x in t[1:]]
        elif t[0] in ("msc", "isc", "tsc", "h"):
            out[t[0]][int(t[1])] = np.array([float.fromhex(x) for x in t[2:]])
        elif t[0] == "maxm":
            out["maxm"][int(t[1])] = float.fromhex(t[2])
    return out


def parse_probe_astar(lines):
    """-> list of dict(idx, kmer, start_state, R=dict, L=dict, contig)"""
    res = []
    for l in lines:
        t = l.split()
        iR, iL, ic = t.index("R"), t.index("L"), t.index("contig")

        def side(tt):
            d = dict(zip(tt[0::2], tt[1::2]))
            return dict(ok=int(d["ok"]), real=float.fromhex(d["real"]), score=float.fromhex(d["score"]), fval=int(d["fval"]),
                        length=int(d["len"]), state_no=int(d["state_no"]), state=d["state"], node=int(d["node"]),
                        closed=int(d["closed"]), seq="" if d["seq"] == "." else d["seq"])
        res.append(dict(idx=int(t[1]), kmer=t[2], start_state=int(t[3]), R=side(t[iR + 1:iL]), L=side(t[iL + 1:ic]),
                        contig=t[ic + 1]))
    return res


def write_buildlib_inputs(d: str) -> str:
    """seeded read files for `megagta buildlib` (multi-line FASTA with empty reads / N / lower case, gzip'ed paired FASTQ,
    interleaved FASTQ whose quality lines start with '@' and '>'); returns the path of the read_lib file"""
    import gzip
    rng = np.random.default_rng(3)

    def seq(n):
        s = "".join("ACGT"[x] for x in rng.integers(0, 4, n))
        if n > 30 and rng.random() < 0.3:
            s = s[:5] + "N" + s[6:20] + "n" + s[21:]
        if rng.random() < 0.2:
            s = s.lower()
        return s

    with open(f"{d}/a.fa", "w") as f:
        for i in range(200):
            s = seq(int(rng.integers(0, 300)))
            f.write(f">r{i} comment\n")
            for j in range(0, len(s), 60):
                f.write(s[j:j + 60] + "\n")
    for tag in "12":
        with gzip.GzipFile(f"{d}/p{tag}.fq.gz", "wb", mtime=0) as f:
            for i in range(150):
                s = seq(int(rng.integers(20, 160)))
                f.write(f"@p{i}/{tag}\n{s}\n+\n{'I' * len(s)}\n".encode())
    with open(f"{d}/i.fq", "w") as f:
        for i in range(100):
            s = seq(100)
            q = ("@" if i % 3 == 0 else ">" if i % 3 == 1 else "I") + "I" * 99
            f.write(f"@i{i}\n{s}\n+i{i}\n{q}\n")
    lib = f"{d}/reads.lib"
    open(lib, "w").write(f"a.fa\nse {d}/a.fa\np1.fq.gz,p2.fq.gz\npe {d}/p1.fq.gz {d}/p2.fq.gz\ni.fq\ninterleaved {d}/i.fq\n")
    return lib


# ---- tests/golden/bigm: models too long for the LDS of a CU.  The inputs are regenerated from their seeds (the committed goldens hold the
# reference's answers and an md5 of every input).
BIGM_CASES = {"m600": dict(M=600, n_reads=2400, seed=61, n_seeds=40), "m1200": dict(M=1200, n_reads=3000, seed=62, n_seeds=32)}


def bigm_inputs(case: str, outdir: str):
    """the seeded inputs of one case: reads (uint8 codes) and gene models written under outdir/genes -> (metagenome, gene dir)"""
    from megagta_amd import synth
    c = BIGM_CASES[case]
    mg = synth.make_metagenome(c["n_reads"], 150, ((case, c["M"]),), seed=c["seed"], reads_per_genome=600, genome_len=6000)
    synth.write_gene_models(mg.genes, os.path.join(outdir, "genes"))
    return mg, os.path.join(outdir, "genes", case)


def bigm_case(golden_dir: str, case: str, outdir: str):
    """-> (packed reads, start_idx, gene dir, cold goldens, warm goldens); checks the regenerated inputs against the stored md5s"""
    import hashlib
    from megagta_amd import synth
    meta = json.load(open(os.path.join(golden_dir, "bigm", "cases.json")))[case]
    mg, gdir = bigm_inputs(case, outdir)
    for tag, name in (("for", "for_enone.hmm"), ("rev", "rev_enone.hmm")):
        assert hashlib.md5(open(os.path.join(gdir, name), "rb").read()).hexdigest() == meta["md5"][tag], f"{case}: regenerated {name} differs from the generator's"
    synth.write_lib_bin(mg.reads, os.path.join(outdir, "reads.lib"))
    assert hashlib.md5(open(os.path.join(outdir, "reads.lib.bin"), "rb").read()).hexdigest() == meta["md5"]["reads"], f"{case}: regenerated reads differ"
    packed, start = synth.pack_reads_for_build(mg.reads)
    gold = {m: parse_probe_astar(gz_lines(os.path.join(golden_dir, "bigm", f"{case}_astar_{m}.txt.gz"))) for m in ("cold", "warm")}
    return packed, start, gdir, gold["cold"], gold["warm"]


def fasta_seqs(path):
    return [l for l in open(path).read().splitlines() if l and l[0] != ">"]


def stagewise_vs_reference(out, work, gene_list: str, oracle, ref_bin: str, our_bin: str, min_multiset: float = 0.95):
    """A finished `megagta.py -k 30,36,45` run under `out` against the reference BINARY fed the same inputs, stage by stage (the
    reference's driver sequence megagta.py:538-720): the three graphs (buildgraph with the previous k's contigs as assist sequences),
    the contigs of the intermediate k (denovo -t 1), the seeds of every gene (findstart, sorted: the reference shuffles its lines),
    and the raw contigs of every gene: MEGAGTA_CACHE_WINDOW=1 byte-identical to the reference's `search ... 1` on the same graph and
    seed files, the driver's own run (whatever window it ran with) compared as a multiset.  -> {gene: (equal as a multiset, seeds)}"""
    return stagewise_finish(stagewise_start(out, work, gene_list, ref_bin, our_bin), oracle, min_multiset)


def stagewise_start(out, work, gene_list: str, ref_bin: str, our_bin: str):
    """the reference's steps of stagewise_vs_reference, started (host threads; only `w1` uses the GPU) -- a caller with GPU work of its own to do
    in the meantime (the two-rank run of the config-4 test) joins them later with stagewise_finish"""
    import subprocess
    from concurrent.futures import ThreadPoolExecutor
    run = lambda cmd, **kw: subprocess.run(cmd, check=True, capture_output=True, **kw)
    lib = str(out / "tmp" / "reads.lib")
    common = ["-m", "1", "--host_mem", "4000000000", "--mem_flag", "1", "--gpu_mem", "0", "--num_cpu_threads", "4", "--num_output_threads", "1",
              "--read_lib_file", lib]
    genes = {l.split()[0]: l.split()[3] for l in open(gene_list)}
    # every reference step takes its inputs from the FINISHED run under `out` (the previous k's contigs, the graph and seed files), so the
    # steps do not wait for each other: they run side by side on the host cores (one after the other they were most of the test's minute)
    jobs = {}
    ex = ThreadPoolExecutor(max_workers=8)
    prev = None
    for k in (29, 35, 44):                                            # the three graphs
        cmd = [ref_bin, "buildgraph", "-k", str(k), "--output_prefix", str(work / f"ref_{k}")] + common
        if prev is not None:
            cmd += ["--assist_seq", str(out / f"k{prev}" / f"{prev}.contigs.fa")]
        jobs[("graph", k)] = ex.submit(run, cmd)
        prev = k
    for k, nxt in ((29, 35), (35, 44)):                               # the contigs of the intermediate k
        jobs[("denovo", k)] = ex.submit(run, [ref_bin, "denovo", "-s", str(out / f"k{k}" / f"{k}"), "-o", str(work / f"ref_{k}"), "-t", "1", "--min_standalone", "400",
                                              "--max_tip_len", "150", "--min_contig", str(nxt + 1)])
    for gene, faa in genes.items():                                   # the seeds of every gene
        jobs[("seeds", gene)] = ex.submit(run, [ref_bin, "findstart", faa, lib + ".bin", "45", "2", str(out / "k35" / "35.contigs.fa")])
    # the reference's one-thread search on OUR graph files and seed files; window 1 == that run, byte for byte, all genes in one call
    jobs["ref1"] = ex.submit(run, [ref_bin, "search", str(out / "k44" / "44"), gene_list, str(out / "k44" / "44"), str(work / "ref1"), "20", "0.5", "1"])
    jobs["w1"] = ex.submit(run, [our_bin, "search", str(out / "k44" / "44"), gene_list, str(out / "k44" / "44"), str(work / "ours_w1"), "20", "0.5", "4"],
                           env={**os.environ, "MEGAGTA_CACHE_WINDOW": "1"})
    return {"ex": ex, "jobs": jobs, "out": out, "work": work, "genes": genes}


def stagewise_finish(started, oracle, min_multiset: float = 0.95):
    from collections import Counter
    out, work, genes = started["out"], started["work"], started["genes"]
    try:
        done = {key: f.result() for key, f in started["jobs"].items()}
    finally:
        started["ex"].shutdown(wait=True)
    for k in (29, 35, 44):
        assert oracle.Stream.read(str(out / f"k{k}" / f"{k}")).edges().md5() == oracle.Stream.read(str(work / f"ref_{k}")).edges().md5(), k
    for k in (29, 35):
        assert (out / f"k{k}" / f"{k}.contigs.fa").read_text() == (work / f"ref_{k}.contigs.fa").read_text(), k
    n_seeds = {}
    for gene in genes:
        ref_seeds = done[("seeds", gene)].stdout.decode().splitlines()
        ours = (out / "k44" / f"44_{gene}_starting_kmers.txt").read_text().splitlines()
        assert ours == sorted(ref_seeds) and len(ours) > 64, gene
        n_seeds[gene] = len(ours)
    res = {}
    for gene in genes:
        assert (work / f"ours_w1_raw_contigs_{gene}.fasta").read_text() == (work / f"ref1_raw_contigs_{gene}.fasta").read_text(), gene
        ours, ref = fasta_seqs(out / "k44" / f"44_raw_contigs_{gene}.fasta"), fasta_seqs(work / f"ref1_raw_contigs_{gene}.fasta")
        assert len(ours) == len(ref) == n_seeds[gene]
        a, b = Counter(ours), Counter(ref)
        common_n = sum((a & b).values())
        assert common_n >= min_multiset * len(ref), (gene, common_n, len(ref))
        res[gene] = (common_n, len(ref), a == b)
        assert (out / "contigs" / gene / "nucl_merged.fasta").stat().st_size > 0 and (out / "contigs" / gene / "prot_merged.fasta").stat().st_size > 0
    return res
