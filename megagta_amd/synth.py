"""Seeded synthetic inputs for the MegaGTA hot path.

The real RDP gene models (share/RDPTools) are absent from the reference snapshot, so every input
is synthetic: a "metagenome" of random 20 kb genomes, each carrying one diverged copy of every
gene, 150 bp reads with substitution errors, HMMER3/b text models accepted by the reference's
`Parser::readHMM` (hmmer3b_parser.h:19-177) and `*_starting_kmers.txt` seed files in the 8-column
layout `search` reads (search.cpp:149-158).  Pure numpy; no reference code involved.

File formats written here (host side of the drop-in boundary):
  * reads.lib.bin / .lib_info  -- sequence_manager.cpp:375-410, read_lib_functions-inl.h:216-225
  * FASTA reads
  * HMMER3 ASCII models (forward + reversed), ref_aligned.faa, gene_list.txt
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np

AA_ORDER = "ACDEFGHIKLMNPQRSTVWY"   # HMMER3 alphabet order on the `HMM` line
DNA = np.frombuffer(b"ACGT", dtype=np.uint8)

# standard genetic code, index = c1*16 + c2*4 + c3 with A0 C1 G2 T3
_CODON_AA = "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSS*CWCLFLF"


def codons_for(aa: str) -> list[int]:
    return [i for i, a in enumerate(_CODON_AA) if a == aa]


@dataclass
class Gene:
    name: str
    M: int
    consensus: str                 # protein, length M
    variants: list[np.ndarray] = field(default_factory=list)   # nucleotide codes (3*M) per genome


@dataclass
class Metagenome:
    reads: np.ndarray              # uint8 [n_reads, L] codes 0..3 (forward orientation as sequenced)
    genes: list[Gene]
    gene_pos: np.ndarray           # [n_genomes, n_genes] offset of the gene copy in its genome
    genome_len: int


def make_metagenome(n_reads: int, read_len: int = 150, gene_specs=(("rplB", 277),), seed: int = 1,
                    genome_len: int = 20000, aa_sub: float = 0.10, err: float = 0.005,
                    reads_per_genome: int = 2000) -> Metagenome:
    """G = n_reads/reads_per_genome random genomes (coverage ~ reads_per_genome*L/genome_len = 15x)."""
    rng = np.random.default_rng(seed)
    n_genomes = max(1, n_reads // reads_per_genome)
    genes: list[Gene] = []
    aa_arr = np.frombuffer(AA_ORDER.encode(), dtype=np.uint8)
    for name, M in gene_specs:
        cons = "".join(AA_ORDER[i] for i in rng.integers(0, 20, size=M))
        genes.append(Gene(name=name, M=M, consensus=cons))
    codon_lists = {a: np.array(codons_for(a)) for a in AA_ORDER}

    genomes = rng.integers(0, 4, size=(n_genomes, genome_len), dtype=np.uint8)
    gene_pos = np.zeros((n_genomes, len(genes)), dtype=np.int64)
    slot = genome_len // (len(genes) + 1)
    for g in range(n_genomes):
        for gi, gene in enumerate(genes):
            prot = list(gene.consensus)
            sub = rng.random(gene.M) < aa_sub
            for i in np.nonzero(sub)[0]:
                prot[i] = AA_ORDER[rng.integers(0, 20)]
            nt = np.empty(3 * gene.M, dtype=np.uint8)
            for i, a in enumerate(prot):
                c = int(rng.choice(codon_lists[a]))
                nt[3 * i] = c >> 4
                nt[3 * i + 1] = (c >> 2) & 3
                nt[3 * i + 2] = c & 3
            pos = gi * slot + int(rng.integers(0, max(1, slot - 3 * gene.M)))
            genomes[g, pos:pos + 3 * gene.M] = nt
            gene_pos[g, gi] = pos
            gene.variants.append(nt)
    # reads: uniform genome / position / strand, substitution errors
    reads = np.empty((n_reads, read_len), dtype=np.uint8)
    chunk = 1 << 18
    ar = np.arange(read_len, dtype=np.int64)
    for s in range(0, n_reads, chunk):
        e = min(n_reads, s + chunk)
        gi = rng.integers(0, n_genomes, size=e - s)
        pos = rng.integers(0, genome_len - read_len + 1, size=e - s)
        r = genomes[gi[:, None], pos[:, None] + ar[None, :]]
        strand = rng.random(e - s) < 0.5
        r[strand] = 3 - r[strand][:, ::-1]
        errs = rng.random(r.shape) < err
        r[errs] = (r[errs] + rng.integers(1, 4, size=int(errs.sum()), dtype=np.uint8)) & 3
        reads[s:e] = r
    return Metagenome(reads=reads, genes=genes, gene_pos=gene_pos, genome_len=genome_len)


def make_strain_mix(seed: int, n_genomes: int = 3, genome_len: int = 3000, read_len: int = 100, cov: int = 20, snp_every: int = 150,
                    err: float = 0.004, tricky: bool = False) -> list[np.ndarray]:
    """reads of n_genomes random genomes, each sequenced together with a second strain (SNPs, a few 1-base indels) at the same depth:
    bubbles whose branches tie in multiplicity, plus the tips and bubbles of substitution errors (the `denovo` test input)"""
    rng = np.random.default_rng(seed)
    seqs = []
    for _ in range(n_genomes):
        g = rng.integers(0, 4, size=genome_len, dtype=np.uint8)
        if tricky:      # hairpins (a stretch followed by its reverse complement), tandem repeats, a stretch shared between genomes
            for _ in range(3):
                p, n = int(rng.integers(100, genome_len - 400)), int(rng.integers(20, 120))
                g[p + n:p + 2 * n] = 3 - g[p:p + n][::-1]
            for _ in range(2):
                p, u, c = int(rng.integers(100, genome_len - 400)), int(rng.integers(5, 40)), int(rng.integers(2, 6))
                g[p:p + u * c] = np.tile(g[p:p + u], c)
            if seqs:
                p, n = int(rng.integers(100, genome_len - 400)), int(rng.integers(60, 300))
                g[p:p + n] = seqs[0][p:p + n]
        s = g.copy()
        grid = np.arange(snp_every // 2, genome_len - 50, snp_every)
        pos = grid + rng.integers(-20, 20, size=grid.size)
        s[pos] = (s[pos] + rng.integers(1, 4, size=pos.size, dtype=np.uint8)) & 3
        s = list(s)
        for p in sorted(rng.integers(200, genome_len - 200, size=3), reverse=True):
            if rng.random() < 0.5:
                del s[p]
            else:
                s.insert(p, int(rng.integers(0, 4)))
        seqs += [g, np.array(s, dtype=np.uint8)]
    reads = []
    for g in seqs:
        n = cov * len(g) // read_len
        for p in rng.integers(0, len(g) - read_len + 1, size=n):
            r = g[p:p + read_len].copy()
            e = rng.random(read_len) < err
            r[e] = (r[e] + rng.integers(1, 4, size=int(e.sum()), dtype=np.uint8)) & 3
            if rng.random() < 0.5:
                r = 3 - r[::-1]
            reads.append(r.astype(np.uint8))
    order = rng.permutation(len(reads))
    return [reads[i] for i in order]


# ----------------------------------------------------------------------------------------------
# 2-bit packing (A0 C1 G2 T3, base j of a word at bits 30-2j: sequence_package.h:126-129)
# ----------------------------------------------------------------------------------------------
def pack_concat(codes: np.ndarray) -> np.ndarray:
    """Pack a flat array of base codes into big-endian-in-word uint32 words (zero padded)."""
    n = codes.size
    nw = (n + 15) // 16
    buf = np.zeros(nw * 16, dtype=np.uint32)
    buf[:n] = codes
    buf = buf.reshape(nw, 16)
    shifts = (30 - 2 * np.arange(16)).astype(np.uint32)
    return np.bitwise_or.reduce(buf << shifts[None, :], axis=1).astype(np.uint32)


def pack_reads_for_build(reads: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """Device-side input of the SdBG build: every read REVERSED (not complemented), concatenated
    (`buildgraph` loads the library with is_reverse=true: cx1_read2sdbg_s1.cpp:97,117).
    Returns (packed uint32 words, start_idx uint64[n+1] in bases)."""
    n, L = reads.shape
    start = (np.arange(n + 1, dtype=np.uint64) * np.uint64(L))
    chunk = 1 << 19                                   # reads per chunk; chunk*L is a multiple of 16
    if n <= chunk:
        return pack_concat(reads[:, ::-1].reshape(-1)), start
    parts = [pack_concat(reads[s:s + chunk, ::-1].reshape(-1)) for s in range(0, n - n % chunk, chunk)]
    tail = reads[n - n % chunk:, ::-1].reshape(-1)
    if tail.size:
        parts.append(pack_concat(tail))
    return np.concatenate(parts), start


def write_lib_bin(reads: np.ndarray, prefix: str, metadata: str = "synthetic.fa") -> None:
    """reads.lib.bin: per read uint32 len + ceil(len/16) words, FORWARD orientation
    (sequence_manager.cpp:375-410); reads.lib.lib_info text (read_lib_functions-inl.h:216-225)."""
    n, L = reads.shape
    wpr = (L + 15) // 16
    buf = np.zeros((n, wpr * 16), dtype=np.uint32)
    buf[:, :L] = reads
    shifts = (30 - 2 * np.arange(16)).astype(np.uint32)
    words = np.bitwise_or.reduce(buf.reshape(n, wpr, 16) << shifts[None, None, :], axis=2).astype(np.uint32)
    rec = np.empty((n, wpr + 1), dtype=np.uint32)
    rec[:, 0] = L
    rec[:, 1:] = words
    rec.tofile(prefix + ".bin")
    with open(prefix + ".lib_info", "w") as f:
        f.write(f"{n * L} {n}\n{metadata}\n0 {n - 1} {L} se\n")


def write_fasta(reads: np.ndarray, path: str) -> None:
    n, L = reads.shape
    seq = DNA[reads]
    with open(path, "wb") as f:
        for i in range(n):
            f.write(b">r%d\n" % i)
            f.write(seq[i].tobytes())
            f.write(b"\n")


# ----------------------------------------------------------------------------------------------
# HMMER3/b text models
# ----------------------------------------------------------------------------------------------
def hmm_text(name: str, protein: str, p_cons: float = 0.81, p_other: float = 0.01) -> str:
    """HMMER3 ASCII model with `p_cons` on the consensus residue, `p_other` elsewhere, uniform
    COMPO, fixed transition rows (values are -ln p, `*` = p 0): the shape SURVEY.md App. D
    verified against Parser::readHMM."""
    M = len(protein)
    nl = lambda p: "%.5f" % (-np.log(p))
    out = ["HMMER3/b [synthetic | megagta_amd.synth]", f"NAME  {name}", f"LENG  {M}", "ALPH  amino",
           "RF    no", "CS    no", "MAP   yes", "STATS LOCAL MSV      -9.0000  0.70000",
           "HMM          " + "        ".join(AA_ORDER),
           "            m->m     m->i     m->d     i->m     i->i     d->m     d->d"]
    uni = "  ".join([nl(0.05)] * 20)
    out.append("  COMPO   " + uni)
    out.append("          " + uni)
    # node 0: B->M1, B->I0, B->D1, I0->M1, I0->I0, then d->m = 0.0, d->d = *
    out.append("          " + "  ".join([nl(0.98), nl(0.01), nl(0.01), nl(0.5), nl(0.5), nl(1.0), "*"]))
    for i, a in enumerate(protein, start=1):
        em = [nl(p_cons) if AA_ORDER[j] == a else nl(p_other) for j in range(20)]
        out.append(f"{i:7d}   " + "  ".join(em) + f"  {i:6d} - -")
        out.append("          " + uni)
        if i < M:
            tr = [nl(0.98), nl(0.01), nl(0.01), nl(0.5), nl(0.5), nl(0.7), nl(0.3)]
        else:
            tr = [nl(0.99), nl(0.01), "*", nl(0.5), nl(0.5), nl(1.0), "*"]
        out.append("          " + "  ".join(tr))
    out.append("//")
    return "\n".join(out) + "\n"


def write_gene_models(genes: list[Gene], outdir: str) -> str:
    """Writes <gene>/for_enone.hmm, rev_enone.hmm, ref_aligned.faa and gene_list.txt; returns its path."""
    os.makedirs(outdir, exist_ok=True)
    lines = []
    for g in genes:
        d = os.path.join(outdir, g.name)
        os.makedirs(d, exist_ok=True)
        fw, rv, faa = (os.path.join(d, n) for n in ("for_enone.hmm", "rev_enone.hmm", "ref_aligned.faa"))
        open(fw, "w").write(hmm_text(g.name, g.consensus))
        open(rv, "w").write(hmm_text(g.name + "_rev", g.consensus[::-1]))
        open(faa, "w").write(f">{g.name}_consensus\n{g.consensus}\n")
        lines.append(f"{g.name} {fw} {rv} {faa}\n")
    gl = os.path.join(outdir, "gene_list.txt")
    open(gl, "w").writelines(lines)
    return gl


def synthetic_seeds(gene: Gene, k_nt: int, n_seeds: int, seed: int = 7) -> list[tuple[str, int]]:
    """(k-mer, 1-based model position) pairs cut in frame from the gene variants: a stand-in for
    `findstart` (fast_kmer_filter.cpp:187) on the GPU box, where the reference is absent."""
    rng = np.random.default_rng(seed)
    kp = k_nt // 3
    out = []
    seen = set()
    tries = 0
    while len(out) < n_seeds and tries < 20 * n_seeds:
        tries += 1
        v = gene.variants[int(rng.integers(0, len(gene.variants)))]
        s = int(rng.integers(0, gene.M - kp + 1))
        kmer = DNA[v[3 * s:3 * s + k_nt]].tobytes().decode()
        if kmer in seen:
            continue
        seen.add(kmer)
        out.append((kmer, s + 1))
    return out


def write_seeds(path: str, seeds: list[tuple[str, int]]) -> None:
    """8-column layout of fast_kmer_filter.cpp:187; `search` uses columns 4 and 8."""
    with open(path, "w") as f:
        for kmer, pos in seeds:
            f.write(f"dump_gene_name\tdump_seq_name\tdump\t{kmer}\ttrue\t1\tx\t{pos}\n")


# ----------------------------------------------------------------------------------------------
# The same kind of read set generated and packed ON THE DEVICE (bench.py: 100 M reads in seconds instead of minutes of numpy).
# torch is only used here, as the random generator + gather + bit packing of the synthetic input; nothing on the product path.
# ----------------------------------------------------------------------------------------------
@dataclass
class DeviceMetagenome:
    packed: "object"               # torch int32 [n_words + 16]: reads REVERSED and concatenated, 2 bits per base (pack_reads_for_build)
    start: "object"                # torch int64 [n_reads + 1]
    n_words: int
    n_reads: int
    read_len: int
    genes: list[Gene]              # variants of the first `keep_variants` genomes only (seeds for the search leg)
    sample_reads: np.ndarray       # the first `host_sample` reads on the host (uint8 codes), for the CPU baseline


def make_metagenome_device(n_reads: int, read_len: int = 150, gene_specs=(("rplB", 277),), seed: int = 1, genome_len: int = 20000,
                           aa_sub: float = 0.10, err: float = 0.005, reads_per_genome: int = 2000, device: str = "cuda",
                           host_sample: int = 1_000_000, keep_variants: int = 4096, chunk: int = 4_000_000, on_chunk=None) -> DeviceMetagenome:
    """G = n_reads / reads_per_genome random genomes, one diverged copy of every gene in each, uniform reads, random strand,
    substitution errors: the model of make_metagenome, drawn by the device's generator (so the reads differ from the
    numpy version's; every rank of a multi-GPU run draws the same ones from the same seed).  `on_chunk(first_read, codes[m, L])`, when
    given, receives every chunk of reads on the host (to write a reads file of any size without holding it)."""
    import torch
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    rng = np.random.default_rng(seed)
    n_genomes = max(1, n_reads // reads_per_genome)
    L = read_len
    genomes = torch.randint(0, 4, (n_genomes, genome_len), dtype=torch.uint8, device=device, generator=gen)
    # codon lists per amino acid (padded to 6) for the random synonymous codons
    ctab = np.zeros((20, 6), dtype=np.int64)
    ccnt = np.zeros(20, dtype=np.int64)
    for ai, a in enumerate(AA_ORDER):
        cs = codons_for(a)
        ctab[ai, :len(cs)] = cs
        ccnt[ai] = len(cs)
    ctab_d, ccnt_d = torch.from_numpy(ctab).to(device), torch.from_numpy(ccnt).to(device)
    genes: list[Gene] = []
    slot = genome_len // (len(gene_specs) + 1)
    ar_g = torch.arange(n_genomes, device=device)
    for gi, (name, M) in enumerate(gene_specs):
        cons_idx = rng.integers(0, 20, size=M)
        gene = Gene(name=name, M=M, consensus="".join(AA_ORDER[i] for i in cons_idx))
        prot = torch.from_numpy(cons_idx).to(device).unsqueeze(0).repeat(n_genomes, 1)                     # [G, M]
        sub = torch.rand((n_genomes, M), device=device, generator=gen) < aa_sub
        prot = torch.where(sub, torch.randint(0, 20, (n_genomes, M), device=device, generator=gen), prot)
        pick = (torch.rand((n_genomes, M), device=device, generator=gen) * ccnt_d[prot]).long().clamp_(max=5)
        codon = ctab_d[prot, torch.minimum(pick, ccnt_d[prot] - 1)]                                         # [G, M] 0..63
        nt = torch.stack([codon >> 4, (codon >> 2) & 3, codon & 3], dim=2).reshape(n_genomes, 3 * M).to(torch.uint8)
        pos = gi * slot + torch.randint(0, max(1, slot - 3 * M), (n_genomes,), device=device, generator=gen)
        cols = pos.unsqueeze(1) + torch.arange(3 * M, device=device).unsqueeze(0)
        genomes[ar_g.unsqueeze(1), cols] = nt
        gene.variants = [v for v in nt[:keep_variants].cpu().numpy()]
        genes.append(gene)
    flat = genomes.reshape(-1)
    n_bases = n_reads * L
    n_words = (n_bases + 15) // 16
    packed = torch.zeros(n_words + 16, dtype=torch.int32, device=device)
    shifts = (30 - 2 * torch.arange(16, device=device, dtype=torch.int32))
    ar_l = torch.arange(L, device=device)
    sample = []
    chunk -= chunk % 8                                               # 8 reads of 150 bases = 75 words: chunks end on word boundaries
    assert (8 * L) % 16 == 0
    for s in range(0, n_reads, chunk):
        e = min(n_reads, s + chunk)
        m = e - s
        g_i = torch.randint(0, n_genomes, (m,), device=device, generator=gen)
        p = torch.randint(0, genome_len - L + 1, (m,), device=device, generator=gen)
        r = flat[(g_i * genome_len + p).unsqueeze(1) + ar_l.unsqueeze(0)]                                   # [m, L] uint8
        strand = torch.rand((m,), device=device, generator=gen) < 0.5
        r = torch.where(strand.unsqueeze(1), 3 - r.flip(1), r)
        errs = torch.rand((m, L), device=device, generator=gen) < err
        r = torch.where(errs, (r + torch.randint(1, 4, (m, L), device=device, generator=gen, dtype=torch.uint8)) & 3, r)
        if len(sample) * chunk < host_sample:
            sample.append(r[: max(0, host_sample - len(sample) * chunk)].cpu().numpy())
        if on_chunk is not None:
            on_chunk(s, r.cpu().numpy())
        rr = r.flip(1).reshape(-1)                                   # reversed, not complemented (cx1_read2sdbg_s1.cpp:97,117)
        pad = (-rr.numel()) % 16
        if pad:
            rr = torch.cat([rr, torch.zeros(pad, dtype=torch.uint8, device=device)])
        w = (rr.view(-1, 16).to(torch.int32) << shifts).sum(dim=1, dtype=torch.int32)
        w0 = s * L // 16
        packed[w0:w0 + w.numel()] = w
        del r, rr, w, errs, g_i, p, strand
    start = torch.arange(n_reads + 1, device=device, dtype=torch.int64) * L
    if str(device).startswith("cuda"):
        torch.cuda.synchronize()
    return DeviceMetagenome(packed=packed, start=start, n_words=n_words, n_reads=n_reads, read_len=L, genes=genes,
                            sample_reads=np.concatenate(sample) if sample else np.zeros((0, L), np.uint8))
