"""Read ingestion on the device: `megagta buildlib` packs the reads with mgta_reads_pack_text; PREFIX.bin /
PREFIX.lib_info byte-identical to the compiled reference's (golden MD5s) and to the host packer's, and the ABI entry against a numpy
restatement on ragged / empty / odd-character reads."""
import ctypes as C
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "megagta_amd", "bin", "megagta")


def test_buildlib_on_the_device_matches_reference_goldens(tmp_path, golden_dir):
    from tests import helpers as H
    assert os.path.exists(BIN), "megagta_amd/bin/megagta missing: run __graft_entry__.build()"
    lib = H.write_buildlib_inputs(str(tmp_path))                      # FASTA + FASTQ + gz, paired / single / interleaved, lengths 0..300, N, lower case
    r = subprocess.run([BIN, "buildlib", lib, str(tmp_path / "dev")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "packed on the device" in r.stderr                         # the kernel ran, not the host loop
    fx = json.load(open(os.path.join(golden_dir, "buildlib.json")))
    data = open(tmp_path / "dev.bin", "rb").read()
    assert len(data) == fx["bin_bytes"] and hashlib.md5(data).hexdigest() == fx["bin_md5"]
    r = subprocess.run([BIN, "buildlib", lib, str(tmp_path / "host")], capture_output=True, text=True, env={**os.environ, "MEGAGTA_BUILDLIB_HOST": "1"})
    assert r.returncode == 0 and "on the host" in r.stderr
    assert open(tmp_path / "host.bin", "rb").read() == data
    assert open(tmp_path / "host.lib_info").read() == open(tmp_path / "dev.lib_info").read()


def test_pack_text_abi_vs_numpy():
    from megagta_amd import api, _lib
    rng = np.random.default_rng(5)
    alphabet = np.frombuffer(b"ACGTacgtNnXx-.", dtype=np.uint8)
    lens = [0, 1, 15, 16, 17, 31, 32, 33, 150, 151, 1024, 1025, 4099, 0, 64] + [int(x) for x in rng.integers(0, 400, 3000)]
    reads = [alphabet[rng.integers(0, alphabet.size, n)] for n in lens]
    text = np.concatenate(reads) if reads else np.zeros(0, np.uint8)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    code = np.zeros(256, np.uint32)
    for ch, v in ((b"Cc", 1), (b"GgNn", 2), (b"Tt", 3)):
        for b in ch:
            code[b] = v
    want = []
    for r in reads:
        n = r.size
        c = np.zeros((n + 15) // 16 * 16, np.uint32)
        c[:n] = code[r]
        w = (c.reshape(-1, 16) << (30 - 2 * np.arange(16, dtype=np.uint32))).sum(axis=1, dtype=np.uint64).astype(np.uint32) if n else np.zeros(0, np.uint32)
        want.append(np.concatenate([[np.uint32(n)], w]))
    want = np.concatenate(want).astype(np.uint32)
    ctx = api.Context(0)
    out = np.zeros(want.size + 8, np.uint32)
    n_words = C.c_uint64()
    rc = ctx._L.mgta_reads_pack_text(ctx.h, text.tobytes(), text.size, off.ctypes.data, len(reads), out.ctypes.data, out.size, C.byref(n_words))
    assert rc == 0, ctx._L.mgta_last_error()
    assert n_words.value == want.size and np.array_equal(out[:want.size], want)
    # too little room and offsets that do not cover the text are refused, nothing is written past the buffer
    rc = ctx._L.mgta_reads_pack_text(ctx.h, text.tobytes(), text.size, off.ctypes.data, len(reads), out.ctypes.data, 10, C.byref(n_words))
    assert rc != 0 and n_words.value == want.size
    bad = off.copy()
    bad[-1] -= 1
    assert ctx._L.mgta_reads_pack_text(ctx.h, text.tobytes(), text.size, bad.ctypes.data, len(reads), out.ctypes.data, out.size, C.byref(n_words)) != 0
    # an empty batch
    z = np.zeros(1, np.uint64)
    assert ctx._L.mgta_reads_pack_text(ctx.h, b"", 0, z.ctypes.data, 0, out.ctypes.data, out.size, C.byref(n_words)) == 0 and n_words.value == 0
    ctx.close()
