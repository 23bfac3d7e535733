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
