"""The PRODUCT's two HMMER3 text parsers -- megagta_amd/hmm.py (Python callers, tests, bench) and csrc/host/formats.cpp::parse_hmm
(`megagta search`) -- against the tables the reference's Parser::readHMM + MostProbablePath produce (hmmer3b_parser.h:19-177,
most_probable_path.h:48-118; hex doubles printed by oracle/_ref/probe, committed under tests/golden/): bit for bit, including `*`
entries (p = 0 -> -inf), a non-uniform COMPO line, a lower-case alphabet letter, and models of 600 / 1200 columns.  A file without a
COMPO line makes the reference read compo[j] of an empty vector (:63-75,122-124: undefined behaviour), so both parsers refuse it loudly."""
import os
import subprocess

import numpy as np
import pytest

from megagta_amd import hmm as hmmlib
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "megagta_amd", "bin", "megagta")

CASES = [("toy/for_enone.hmm", "toy/hmm_for.txt.gz"), ("toy/rev_enone.hmm", "toy/hmm_rev.txt.gz"), ("bigm/star.hmm", "bigm/star_tables.txt.gz")]


def _same(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))      # bit for bit (-inf == -inf, no NaN in these tables)


def _check_tables(got, ref):
    """got: dict(M, A, alpha, msc[k], tsc[k], maxm[k], h[k]) in the probe's layout"""
    assert (got["M"], got["A"]) == (ref["M"], ref["A"])
    assert list(got["alpha"]) == ref["alpha"]
    n_inf = 0
    for k in range(ref["M"] + 1):
        if k > 0:                                                   # node 0 has no match line (msc(0, .) is never read: profile_hmm.h:58-64)
            assert _same(got["msc"][k], ref["msc"][k]), k
            n_inf += int(np.isinf(ref["msc"][k]).sum())
        assert _same(got["tsc"][k], ref["tsc"][k]), k
        n_inf += int(np.isinf(ref["tsc"][k]).sum())
        assert _same([got["maxm"][k]], [ref["maxm"][k]]), k
        assert _same(got["h"][k], ref["h"][k]), k
    return n_inf


@pytest.mark.parametrize("model,tables", CASES)
def test_python_parser_tables_vs_reference(golden_dir, model, tables):
    hm = hmmlib.parse_hmm(os.path.join(golden_dir, model))
    ref = H.parse_probe_hmm(H.gz_lines(os.path.join(golden_dir, tables)))
    got = dict(M=hm.M, A=hm.A, alpha=hm.alpha, msc=hm.msc, tsc=hm.tsc.T, maxm=hm.max_match, h=hm.h.T)
    n_inf = _check_tables(got, ref)
    for k in range(hm.M + 1):                                       # insert emissions: 0 in normalized mode, -inf at node M (:145-147,170-172)
        assert _same(hm.isc[k], ref["isc"][k])
    if "star" in model:
        assert n_inf >= 30                                          # the `*` entries are there and were compared


@pytest.mark.parametrize("model,tables", CASES)
def test_cpp_parser_tables_vs_reference(golden_dir, model, tables):
    if not os.path.exists(BIN):
        pytest.skip("megagta_amd/bin/megagta is not built")
    out = subprocess.run([BIN, "hmmdump", os.path.join(golden_dir, model)], check=True, capture_output=True, text=True).stdout
    got = H.parse_probe_hmm(out.splitlines())
    ref = H.parse_probe_hmm(H.gz_lines(os.path.join(golden_dir, tables)))
    _check_tables(got, ref)


@pytest.mark.parametrize("case", sorted(H.BIGM_CASES))
def test_parsers_agree_on_models_beyond_the_lds(golden_dir, tmp_path, case):
    """600 / 1200 columns (the files are regenerated from their seeds and checked against the md5 of the ones the reference's goldens were
    made from): the two product parsers produce the same bits -- what reaches the device does not depend on which front end parsed it"""
    if not os.path.exists(BIN):
        pytest.skip("megagta_amd/bin/megagta is not built")
    _, gdir = H.bigm_inputs(case, str(tmp_path))
    for name in ("for_enone.hmm", "rev_enone.hmm"):
        hm = hmmlib.parse_hmm(os.path.join(gdir, name))
        out = subprocess.run([BIN, "hmmdump", os.path.join(gdir, name)], check=True, capture_output=True, text=True).stdout
        cpp = H.parse_probe_hmm(out.splitlines())
        assert hm.M == H.BIGM_CASES[case]["M"]
        _check_tables(dict(M=hm.M, A=hm.A, alpha=hm.alpha, msc=hm.msc, tsc=hm.tsc.T, maxm=hm.max_match, h=hm.h.T), cpp)


def test_missing_compo_line_is_refused(golden_dir, tmp_path):
    lines = open(os.path.join(golden_dir, "bigm", "star.hmm")).read().split("\n")
    assert lines[6].split()[0] == "COMPO"
    bad = str(tmp_path / "nocompo.hmm")
    open(bad, "w").write("\n".join(lines[:6] + lines[7:]))
    with pytest.raises(ValueError, match="COMPO"):
        hmmlib.parse_hmm(bad)
    if os.path.exists(BIN):
        r = subprocess.run([BIN, "hmmdump", bad], capture_output=True, text=True)
        assert r.returncode != 0 and "COMPO" in r.stderr
