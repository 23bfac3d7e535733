"""CPU restatement of `megagta findstart` (seed finder).   *** TEST INFRASTRUCTURE, NOT PRODUCT CODE ***

Follows fast_kmer_filter.cpp:49-217 (find_start, ProcessSequenceMulti), prot_kmer_generator.h:14-146 (k-mer generator incl. its
model-only rules), prot_kmer.h:27-43 (alphabet) and kmer.h:66-84,149-170 (packing, decodePacked).  Pure Python: small inputs only.
Pinned by tests/test_findstart.py against the compiled reference's output (tests/golden/toy/44_rplB_starting_kmers.txt,
tests/golden/findstart/*), as a sorted multiset of lines: the reference shuffles its output (`random_shuffle`, :183).
"""
from __future__ import annotations

AA_UPPER = "ARNDCQEGHILKMFPSTWYV"           # prot_kmer.h:31-40: codes 0..19, '*' = 20, everything else 31 (invalid)
CODE = {c: i for i, c in enumerate(AA_UPPER)}
CODE.update({c.lower(): i for i, c in enumerate(AA_UPPER)})
CODE["*"] = 20
DECODE = AA_UPPER.lower() + "*"             # int_to_char: lower case

# standard genetic code, index = 16*b0 + 4*b1 + b2 with A0 C1 G2 T3 (seq::AASequence::translate on unambiguous codons)
CODON_AA = "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSS*CWCLFLF"


def model_kmers(seq: str, kaa: int):
    """ProtKmerGenerator(seq, kaa, model_only=true): yields (k-mer as upper-case string, model position).
    Lower case, '-', 'X', 'x' break the window ('-' and 'X' also occupy a model column); '.', '*' and letters outside the
    alphabet are skipped without breaking it (prot_kmer_generator.h:60-135)."""
    position, klength, window = 1, 0, []
    for base in seq:
        if base.islower() or base in "-X":
            if base in "-X":
                position += 1
            klength = 0
            continue
        if base != "." and base != "*" and base in CODE:
            window.append(base)
            position += 1
            klength += 1
            if klength >= kaa:
                yield "".join(window[-kaa:]), position - kaa
    return


def read_fasta(path: str):
    name, chunks = None, []
    for line in open(path):
        line = line.rstrip("\r\n")
        if line.startswith(">"):
            if name is not None:
                yield name, "".join(chunks)
            name, chunks = line[1:], []
        elif name is not None:
            chunks.append(line.strip())
    if name is not None:
        yield name, "".join(chunks)


def reference_set(faa_path: str, kaa: int) -> dict[tuple[int, ...], int]:
    """the k-mer set of find_start (:81-91): first insertion wins (insert_unique), key = amino-acid codes"""
    ref: dict[tuple[int, ...], int] = {}
    for _, seq in read_fasta(faa_path):
        for kmer, pos in model_kmers(seq, kaa):
            ref.setdefault(tuple(CODE[c] for c in kmer), pos)
    return ref


def revcomp(s: str) -> str:
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


def seeds_of_sequence(seq: str, ref: dict, k: int):
    """ProcessSequenceMulti (:193-215): three frames of one strand; every window of k/3 residues that is in the set"""
    kaa = k // 3
    idx = {"A": 0, "C": 1, "G": 2, "T": 3}
    for gen in range(3):
        n_aa = (len(seq) - gen) // 3
        aa = [CODE[CODON_AA[16 * idx[seq[gen + 3 * i]] + 4 * idx[seq[gen + 3 * i + 1]] + idx[seq[gen + 3 * i + 2]]]] for i in range(n_aa)]
        for a in range(n_aa - kaa + 1):
            key = tuple(aa[a:a + kaa])
            if key in ref:
                pos = 3 * a + gen
                yield seq[pos:pos + k], "".join(DECODE[c] for c in key), ref[key]


def find_start(faa_path: str, reads: list[str], k: int, contigs: list[str] = ()) -> list[str]:
    """the lines `megagta findstart` prints, sorted (unique by nucleotide k-mer, :181-182)"""
    ref = reference_set(faa_path, k // 3)
    seen = {}
    for s in list(reads) + list(contigs):
        if len(s) < k:                                    # :120,153
            continue
        for strand in (s, revcomp(s)):
            for nucl, prot, pos in seeds_of_sequence(strand, ref, k):
                seen.setdefault(nucl, (prot, pos))
    return sorted(f"dump_gene_name\tdump_seq_name\tdump\t{n}\ttrue\t1\t{p}\t{m}" for n, (p, m) in seen.items())
