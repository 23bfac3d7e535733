"""`megagta buildlib` (host only: file formats): PREFIX.bin / PREFIX.lib_info byte-identical to what the
compiled reference writes (build_read_lib.cpp, read_lib_functions-inl.h:116-225, sequence_manager.cpp:109-216,375-410)."""
import hashlib
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "megagta_amd", "bin", "megagta")
REF = os.path.join(ROOT, "oracle", "_ref", "megagta")


def test_buildlib_matches_reference(tmp_path, golden_dir):
    from tests import helpers as H
    assert os.path.exists(BIN), "megagta_amd/bin/megagta missing: run __graft_entry__.build()"
    lib = H.write_buildlib_inputs(str(tmp_path))
    r = subprocess.run([BIN, "buildlib", lib, str(tmp_path / "out")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    fx = json.load(open(os.path.join(golden_dir, "buildlib.json")))
    data = open(tmp_path / "out.bin", "rb").read()
    assert len(data) == fx["bin_bytes"] and hashlib.md5(data).hexdigest() == fx["bin_md5"]
    info = open(tmp_path / "out.lib_info").read().replace(str(tmp_path), "<d>")
    assert info.splitlines()[0] == fx["lib_info"].splitlines()[0]                      # total bases, total reads
    assert [l for l in info.splitlines() if " " in l and l.split()[-1] in ("se", "pe")] == \
           [l for l in fx["lib_info"].splitlines() if " " in l and l.split()[-1] in ("se", "pe")]
    if os.path.exists(REF):                                                            # and directly, byte for byte
        subprocess.run([REF, "buildlib", lib, str(tmp_path / "ref")], check=True, capture_output=True)
        assert open(tmp_path / "ref.bin", "rb").read() == data
        assert open(tmp_path / "ref.lib_info").read() == open(tmp_path / "out.lib_info").read()
    # odd paired library / unknown type are refused like the reference does (read_lib_functions-inl.h:177-200)
    open(tmp_path / "bad.lib", "w").write(f"x\nweird {tmp_path}/a.fa\n")
    r = subprocess.run([BIN, "buildlib", str(tmp_path / "bad.lib"), str(tmp_path / "bad")], capture_output=True, text=True)
    assert r.returncode != 0 and "Valid types" in r.stderr
    r = subprocess.run([BIN, "buildlib"], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage" in r.stderr


def test_buildlib_reader_across_buffer_refills(tmp_path):
    """the FASTA/FASTQ reader cuts lines out of 1 MiB refills: a file of several refills with CRLF line ends and no final newline gives
    the records the format defines (uint32 length + ceil(len / 16) words, base j of a word at bits 30 - 2j)"""
    import numpy as np
    rng = np.random.default_rng(9)
    n, L = 30_000, 101
    codes = rng.integers(0, 4, (n, L)).astype(np.uint8)
    text = "\r\n".join(f">r{i}\r\n" + "".join("ACGT"[c] for c in codes[i]) for i in range(n))       # ~3.4 MB, no newline at the end
    open(tmp_path / "big.fa", "w", newline="").write(text)
    open(tmp_path / "big.lib", "w").write(f"big\nse {tmp_path}/big.fa\n")
    r = subprocess.run([BIN, "buildlib", str(tmp_path / "big.lib"), str(tmp_path / "big")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    nw = (L + 15) // 16
    padded = np.zeros((n, nw * 16), np.uint32)
    padded[:, :L] = codes
    words = (padded.reshape(n, nw, 16) << (30 - 2 * np.arange(16, dtype=np.uint32))).sum(axis=2, dtype=np.uint32)
    rec = np.empty((n, nw + 1), np.uint32)
    rec[:, 0] = L
    rec[:, 1:] = words
    assert open(tmp_path / "big.bin", "rb").read() == rec.tobytes()
    assert open(tmp_path / "big.lib_info").read().splitlines()[0] == f"{n * L} {n}"


def _libdump(tmp_path, prefix, mode, assist=None, env=None):
    import numpy as np
    out = str(tmp_path / ("dump_" + mode + ("_a" if assist else "")))
    cmd = [BIN, "libdump", prefix, mode, out] + ([assist] if assist else [])
    r = subprocess.run(cmd, capture_output=True, text=True, env={**os.environ, **(env or {})})
    assert r.returncode == 0, r.stderr
    n_reads, n_words, max_len, n_short = (int(x) for x in r.stdout.split())
    w, s = np.fromfile(out + ".words", np.uint32), np.fromfile(out + ".start", np.uint64)
    assert w.size == n_words and s.size == n_reads + 1
    return w, s, max_len, n_short


def test_host_read_loaders_match_python_packers(tmp_path, golden_dir):
    """what `megagta buildgraph` / `findstart` upload (reads reversed, 2 bit/base, concatenated; loaded 16 bases at a time) equals what the
    Python packers give the GPU tests: ragged lengths, empty reads, the bare .bin reader, an assist FASTA appended base by base"""
    import numpy as np
    from megagta_amd import readlib
    from tests import helpers as H
    prefixes = [os.path.join(golden_dir, "toy", "reads.lib"), os.path.join(golden_dir, "ragged", "reads.lib")]
    lib = H.write_buildlib_inputs(str(tmp_path))                      # lengths 0..300, N, lower case
    r = subprocess.run([BIN, "buildlib", lib, str(tmp_path / "mixed")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    prefixes.append(str(tmp_path / "mixed"))
    for prefix in prefixes:
        reads = readlib.load_lib_bin(prefix)
        packed, start = readlib.pack_for_build(reads)
        for mode in ("lib", "bin"):
            w, s, max_len, n_short = _libdump(tmp_path, prefix, mode)
            assert np.array_equal(s, start) and np.array_equal(w, packed), (prefix, mode)
            assert max_len == max(r.size for r in reads) and n_short == len(reads)
    # assist sequences (buildgraph --assist_seq): appended after the library, through the FASTA reader
    rng = np.random.default_rng(2)
    assist = [rng.integers(0, 4, int(n)).astype(np.uint8) for n in (37, 500, 1, 64, 129)]
    with open(tmp_path / "assist.fa", "w") as f:
        for i, a in enumerate(assist):
            f.write(f">c{i}\n" + "".join("ACGT"[c] for c in a) + "\n")
    open(tmp_path / "assist.fa.info", "w").write(f"{len(assist)} {sum(a.size for a in assist)}\n")
    reads = readlib.load_lib_bin(prefixes[1])
    packed, start = readlib.pack_for_build(reads + assist)
    w, s, _, n_short = _libdump(tmp_path, prefixes[1], "lib", str(tmp_path / "assist.fa"))
    assert np.array_equal(s, start) and np.array_equal(w, packed) and n_short == len(reads)
    # the worker's route for the contigs its own `denovo` has just made: the FASTA text in memory, records found and packed by all host
    # threads (PackedReads::append_text_many) -- the same words whatever the number of threads and wherever their pieces of the text begin
    many = [rng.integers(0, 4, int(n)).astype(np.uint8) for n in rng.integers(1, 400, 3000)]
    with open(tmp_path / "many.fa", "w") as f:
        for i, a in enumerate(many):
            f.write(f">k29_{i + 1} flag=1 multi=2.0000 len={a.size}\n" + "".join("ACGT"[c] for c in a) + "\n")
    open(tmp_path / "many.fa.info", "w").write(f"{len(many)} {sum(a.size for a in many)}\n")
    packed, start = readlib.pack_for_build(reads + many)
    for threads in ("1", "3", "8"):
        for route in ("0", "1"):
            w, s, _, n_short = _libdump(tmp_path, prefixes[1], "lib", str(tmp_path / "many.fa"), env={"MEGAGTA_LIBDUMP_TEXT": route, "OMP_NUM_THREADS": threads})
            assert np.array_equal(s, start) and np.array_equal(w, packed) and n_short == len(reads), (threads, route)


def test_host_read_loaders_keep_empty_reads(tmp_path):
    """records of length 0 (buildlib keeps them: sequence_manager.cpp:375-410 writes `len = 0` and no word) through both host loaders,
    reversed as `buildgraph` / `findstart` load them: first, in the middle, several in a row, last (advisor r5: `nw - 1` wrapped)"""
    import numpy as np
    from megagta_amd import readlib
    rng = np.random.default_rng(11)
    lens = [0, 17, 0, 0, 16, 1, 0, 33, 0]
    reads = [rng.integers(0, 4, n).astype(np.uint8) for n in lens]
    with open(tmp_path / "e.fa", "w") as f:
        for i, a in enumerate(reads):
            f.write(f">e{i}\n" + "".join("ACGT"[c] for c in a) + "\n")
    open(tmp_path / "e.lib", "w").write(f"e\nse {tmp_path}/e.fa\n")
    r = subprocess.run([BIN, "buildlib", str(tmp_path / "e.lib"), str(tmp_path / "e")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert open(tmp_path / "e.lib_info").read().splitlines()[0] == f"{sum(lens)} {len(lens)}"
    got = readlib.load_lib_bin(str(tmp_path / "e"))
    assert [g.size for g in got] == lens and all(np.array_equal(a, b) for a, b in zip(got, reads))
    packed, start = readlib.pack_for_build(reads)
    for mode in ("lib", "bin"):
        for threads in ("1", "4"):
            w, s, max_len, n_short = _libdump(tmp_path, str(tmp_path / "e"), mode, env={"OMP_NUM_THREADS": threads})
            assert np.array_equal(s, start) and np.array_equal(w, packed), (mode, threads)
            assert max_len == 33 and n_short == len(lens)


def test_filterbylen_and_translate_match_reference(tmp_path):
    """the driver's two text filters (host only): same bytes as the reference binary on multi-line records, CRLF, comments, a record
    shorter than the limit, lengths not divisible by three, no newline at the end"""
    import numpy as np
    import pytest
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/megagta not built")
    rng = np.random.default_rng(4)
    parts = []
    for i in range(400):
        L = int(rng.integers(1, 1400))
        s = "".join("acgtACGTn"[c] for c in rng.integers(0, 9 if i % 17 == 0 else 4, L))
        head = f">g_contig_{2 * i}_contig_{2 * i + 1}" + (" some comment" if i % 5 == 0 else "")
        eol = "\r\n" if i % 7 == 0 else "\n"
        body = eol.join(s[j:j + 70] for j in range(0, L, 70)) if i % 3 == 0 else s
        parts.append(head + eol + body + eol)
    text = "".join(parts)
    open(tmp_path / "c.fa", "w", newline="").write(text[:-1])            # no newline at the end
    outs = {}
    for tag, exe in (("ours", BIN), ("ref", REF)):
        with open(tmp_path / "c.fa", "rb") as fin:
            r = subprocess.run([exe, "filterbylen", "450"], stdin=fin, capture_output=True)
        assert r.returncode == 0, r.stderr
        open(tmp_path / f"f_{tag}.fa", "wb").write(r.stdout)
        t = subprocess.run([exe, "translate", str(tmp_path / f"f_{tag}.fa")], capture_output=True)
        assert t.returncode == 0, t.stderr
        outs[tag] = (r.stdout, t.stdout)
    assert outs["ours"][0] == outs["ref"][0] and len(outs["ours"][0]) > 10_000
    assert outs["ours"][1] == outs["ref"][1] and len(outs["ours"][1]) > 3_000
