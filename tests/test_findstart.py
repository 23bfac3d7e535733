"""`findstart`: the oracle's restatement against the compiled reference's output (CPU), and the HIP seed
finder against both (GPU).  The reference shuffles its lines, so parity = the sorted list of lines."""
import gzip
import os

import pytest

from megagta_amd import readlib


def _lines(path):
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rt") as f:
        return sorted(l.rstrip("\n") for l in f if l.strip())


def _read_strings(prefix):
    return ["".join("ACGT"[x] for x in r) for r in readlib.load_lib_bin(prefix)]


def _contigs(path):
    from oracle import findstart_oracle as F
    return [s.upper().replace("N", "G") for _, s in F.read_fasta(path)]


def test_oracle_toy_vs_reference(golden_dir):
    from oracle import findstart_oracle as F
    toy = os.path.join(golden_dir, "toy")
    got = F.find_start(os.path.join(toy, "ref_aligned.faa"), _read_strings(os.path.join(toy, "reads.lib")), 45)
    assert got == _lines(os.path.join(toy, "44_rplB_starting_kmers.txt")) and len(got) == 93


@pytest.mark.parametrize("k", [30, 45, 72])
@pytest.mark.parametrize("with_contigs", [False, True])
def test_oracle_quirks_vs_reference(golden_dir, k, with_contigs):
    """model-only generator rules (lower case, '-', '.', 'X', '*', foreign letters, short sequence, first insert wins), both strands,
    reads shorter than k, N -> G, one- and two-word k-mers (k/3 = 10, 15, 24), the optional contig FASTA"""
    from oracle import findstart_oracle as F
    d = os.path.join(golden_dir, "findstart")
    contigs = _contigs(os.path.join(d, "contigs.fa")) if with_contigs else []
    got = F.find_start(os.path.join(d, "ref_quirks.faa"), _read_strings(os.path.join(d, "reads.lib")), k, contigs)
    want = _lines(os.path.join(d, f"seeds_k{k}{'_contigs' if with_contigs else ''}.txt.gz"))
    assert len(want) > 500 and got == want


# ---------------------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def ctx():
    from megagta_amd import api
    c = api.Context(0)
    yield c
    c.close()


@pytest.mark.gpu
def test_gpu_toy_vs_reference(ctx, golden_dir):
    from megagta_amd import findstart
    toy = os.path.join(golden_dir, "toy")
    lines, st = findstart.find_start(ctx, os.path.join(toy, "ref_aligned.faa"), readlib.load_lib_bin(os.path.join(toy, "reads.lib")), 45)
    assert lines == _lines(os.path.join(toy, "44_rplB_starting_kmers.txt"))
    assert st["n_hits"] >= st["n_seeds"] == 93


@pytest.mark.gpu
@pytest.mark.parametrize("k", [30, 45, 72])
@pytest.mark.parametrize("with_contigs", [False, True])
def test_gpu_quirks_vs_reference_and_oracle(ctx, golden_dir, k, with_contigs):
    import numpy as np
    from megagta_amd import findstart
    from oracle import findstart_oracle as F
    d = os.path.join(golden_dir, "findstart")
    reads = readlib.load_lib_bin(os.path.join(d, "reads.lib"))
    contigs = [np.array(["ACGT".index(c) for c in s], dtype=np.uint8) for s in _contigs(os.path.join(d, "contigs.fa"))] if with_contigs else []
    lines, st = findstart.find_start(ctx, os.path.join(d, "ref_quirks.faa"), reads, k, contigs)
    assert lines == _lines(os.path.join(d, f"seeds_k{k}{'_contigs' if with_contigs else ''}.txt.gz"))
    assert lines == F.find_start(os.path.join(d, "ref_quirks.faa"), _read_strings(os.path.join(d, "reads.lib")), k,
                                 _contigs(os.path.join(d, "contigs.fa")) if with_contigs else [])


@pytest.mark.gpu
def test_gpu_forward_and_reversed_storage_agree(ctx, golden_dir):
    """the same hits whether the reads were uploaded forward or reversed (as buildgraph wants them); bad k is refused"""
    import numpy as np
    from megagta_amd import api, findstart
    d = os.path.join(golden_dir, "findstart")
    reads = readlib.load_lib_bin(os.path.join(d, "reads.lib"))
    words, mpos = findstart.reference_words(os.path.join(d, "ref_quirks.faa"), 15)
    pw = findstart.pack_words(words, 15)
    packed_r, start = readlib.pack_for_build(reads)
    packed_f, _ = readlib.pack_for_build([r[::-1] for r in reads])           # reversing twice = forward storage
    h_r, _ = findstart.find_hits(ctx, ctx.upload_reads(packed_r, start), True, 45, pw)
    h_f, _ = findstart.find_hits(ctx, ctx.upload_reads(packed_f, start), False, 45, pw)
    assert h_r.size > 1000 and np.array_equal(np.sort(h_r, order=["read", "pos_strand", "ref"]), np.sort(h_f, order=["read", "pos_strand", "ref"]))
    with pytest.raises(api.MegaGtaError):
        findstart.find_hits(ctx, ctx.upload_reads(packed_r, start), True, 44, pw)
    with pytest.raises(api.MegaGtaError):
        findstart.find_hits(ctx, ctx.upload_reads(packed_r, start), True, 75, pw)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(16)))
def test_gpu_fuzz_vs_oracle(ctx, tmp_path, seed):
    """random alignments over the generator's whole input alphabet, every k/3 in [3, 24], ragged reads on both strands"""
    import numpy as np
    from megagta_amd import findstart
    from oracle import findstart_oracle as F
    rng = np.random.default_rng(500 + seed)
    kaa = int(rng.integers(3, 25))
    alphabet = list("ARNDCQEGHILKMFPSTWYV" * 6 + "arndcq" + "--..XX**BZUxj")
    faa = tmp_path / "ref.faa"
    prots = []
    with open(faa, "w") as f:
        for i in range(int(rng.integers(1, 5))):
            s = "".join(alphabet[int(x)] for x in rng.integers(0, len(alphabet), int(rng.integers(kaa, 200))))
            prots.append(s)
            f.write(f">s{i} d\n")
            for j in range(0, len(s), 37):
                f.write(s[j:j + 37] + "\n")
    codons = {a: [i for i in range(64) if F.CODON_AA[i] == a] for a in F.AA_UPPER + "*"}
    clean = ["".join(c for c in p if c in F.AA_UPPER) for p in prots]
    reads = []
    for _ in range(300):
        L = int(rng.integers(1, 260))
        r = rng.integers(0, 4, L).astype(np.uint8)
        src = clean[int(rng.integers(0, len(clean)))]
        if len(src) >= 2 and rng.random() < 0.8:
            n_aa = int(rng.integers(1, min(len(src), 60) + 1))
            st = int(rng.integers(0, len(src) - n_aa + 1))
            core = []
            for a in src[st:st + n_aa]:
                c = int(rng.choice(codons[a]))
                core += [c >> 4, (c >> 2) & 3, c & 3]
            core = np.array(core[:L], dtype=np.uint8)
            off = int(rng.integers(0, L - core.size + 1))
            r[off:off + core.size] = core
        if rng.random() < 0.5:
            r = (3 - r[::-1]).astype(np.uint8)
        reads.append(r)
    k = 3 * kaa
    lines, st = findstart.find_start(ctx, str(faa), reads, k)
    assert lines == F.find_start(str(faa), ["".join("ACGT"[x] for x in r) for r in reads], k)
