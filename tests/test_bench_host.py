"""bench.py's host logic that no GPU is needed for: the reads -> contigs leg keeps to its time limits.  (The driver's call of bench.py ends
at 600 s; a bench line that never prints is worth less than one without its last optional leg.)"""
import os
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

STAND_IN = r'''
import os, sys, time
a = sys.argv
out = a[a.index("-o") + 1]
genes = [l.split()[0] for l in open(a[a.index("-g") + 1])]
if "--bin" in a:
    time.sleep(float(os.environ.get("STAND_IN_REF_SECONDS", "0")))
for g in genes:
    os.makedirs(os.path.join(out, "contigs", g), exist_ok=True)
    # (the stand-in "reference" writes one contig of its own: three of its four are the ones "ours" writes -- twice ACGT counts twice)
    extra = ">c2\nTTTT\n>c3\nACGT\n" if "--bin" in a and os.environ.get("STAND_IN_REF_EXTRA") else ""
    open(os.path.join(out, "contigs", g, "nucl_merged.fasta"), "w").write(">c0\nACGT\n>c1\nacgt\n" + extra)
'''


def _stand_ins(monkeypatch, tmp_path):
    """the leg with a stand-in for megagta.py (writes two contigs per gene; sleeps when it is asked to run the reference binary) and for the
    read generator (the real one needs a GPU)"""
    import bench
    from megagta_amd import synth
    drv = tmp_path / "driver.py"
    drv.write_text(STAND_IN)
    monkeypatch.setattr(bench, "DRIVER", str(drv))
    monkeypatch.setattr(bench, "REF", sys.executable)                   # (only its existence is looked at; the stand-in never runs it)

    def fake_reads(n, L, specs, seed=0, device=None, host_sample=0):
        return types.SimpleNamespace(genes=[types.SimpleNamespace(name=s[0]) for s in specs], sample_reads=np.zeros((4, L), dtype=np.uint8))

    def fake_models(genes, d):
        os.makedirs(d, exist_ok=True)
        p = os.path.join(d, "gene_list.txt")
        open(p, "w").write("".join(f"{g.name} f r a\n" for g in genes))
        return p

    monkeypatch.setattr(synth, "make_metagenome_device", fake_reads)
    monkeypatch.setattr(synth, "write_gene_models", fake_models)
    return bench


def test_e2e_leg_in_time_measures_every_run(monkeypatch, tmp_path):
    bench = _stand_ins(monkeypatch, tmp_path)
    out = bench.e2e_leg((("rplB", 10), ("nirK", 12)), 100, 50, "cpu", n_large=200, large_deadline=time.time() + 60, hard_stop=time.time() + 60)
    assert out["ours"]["contigs"] == {"rplB": 2, "nirK": 2} and out["reference"]["contigs"] == {"rplB": 2, "nirK": 2}
    assert set(out["reference_thread_sweep"]["seconds_by_threads"]) <= {16, 32, os.cpu_count()} and out["speedup_same_sample"] > 0
    assert out["ours_large"]["reads"] == 200 and "seconds" in out["ours_large"]


def test_e2e_leg_skips_the_large_run_when_late_and_ends_a_run_at_the_hard_stop(monkeypatch, tmp_path):
    bench = _stand_ins(monkeypatch, tmp_path)
    # late for the optional run: it is not started, everything else is measured
    out = bench.e2e_leg((("rplB", 10),), 100, 50, "cpu", n_large=200, large_deadline=time.time() - 1, hard_stop=time.time() + 60)
    assert "skipped" in out["ours_large"] and "seconds" in out["reference"]
    # the reference takes longer than the run has: its process group is ended, what was measured before stays, the leg returns
    monkeypatch.setenv("STAND_IN_REF_SECONDS", "120")
    t = time.time()
    out = bench.e2e_leg((("rplB", 10),), 100, 50, "cpu", n_large=200, large_deadline=time.time() + 60, hard_stop=time.time() + 6)
    assert time.time() - t < 30
    assert "seconds" in out["ours"] and "seconds" in out["ours_unordered_cache"]
    assert "cut_off" in out["reference"] and "ended, not measured" in out["reference"]["cut_off"]
    assert "ours_large" not in out and "speedup_same_sample" not in out


def test_e2e_leg_compares_the_contigs_of_the_two_runs_not_their_number(monkeypatch, tmp_path):
    bench = _stand_ins(monkeypatch, tmp_path)
    out = bench.e2e_leg((("rplB", 10), ("nirK", 12)), 100, 50, "cpu", hard_stop=time.time() + 60)
    assert out["contigs_equal_fraction"] == {"rplB": 1.0, "nirK": 1.0}                  # same sequences (case does not count)
    monkeypatch.setenv("STAND_IN_REF_EXTRA", "1")
    out = bench.e2e_leg((("rplB", 10),), 100, 50, "cpu", hard_stop=time.time() + 60)
    # the reference holds ACGT x 3 + TTTT; ours ACGT x 2: two of its four contigs are matched (a multiset, not a set)
    assert out["reference"]["contigs"] == {"rplB": 4} and out["contigs_equal_fraction"] == {"rplB": 0.5}
    assert out["contigs_equal_fraction_ours_ordered_vs_unordered"] == {"rplB": 1.0}


def test_search_cpu_baseline_keeps_the_thread_counts_that_survive(monkeypatch, tmp_path):
    """the reference's `search` can die of its own race (unlocked find against a rehash): the run that crashed costs its own number only"""
    import subprocess
    import bench
    from megagta_amd import synth
    ref = tmp_path / "ref.py"
    ref.write_text(r"""#!%s
import os, signal, sys
a = sys.argv[1:]
if a[0] == "findstart":
    for i in range(40):
        print("dump_gene_name\tdump_seq_name\tdump\t" + "ACGT" * 11 + "A\ttrue\t1\tx\t%%d" %% (i + 1))
elif a[0] == "search":
    if a[-1] == "16":
        os.kill(os.getpid(), signal.SIGSEGV)
    sys.stderr.write("Done rplB: time %%s\n" %% {"12": "2.0", "32": "1.0"}.get(a[-1], "3.0"))
""" % sys.executable)
    ref.chmod(0o755)
    probe = tmp_path / "probe"
    probe.write_text("#!/bin/sh\necho 'seed 0 closed 100 x'\necho 'seed 1 closed 50 x'\n")
    probe.chmod(0o755)
    monkeypatch.setattr(bench, "REF", str(ref))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "oracle" / "_ref")
    os.replace(probe, tmp_path / "oracle" / "_ref" / "probe")

    def fake_models(genes, d):
        os.makedirs(d, exist_ok=True)
        p = os.path.join(d, "gene_list.txt")
        open(p, "w").write("rplB f r a\n")
        return p

    monkeypatch.setattr(synth, "write_gene_models", fake_models)
    monkeypatch.setattr(os, "cpu_count", lambda: 64)
    sb = bench.search_cpu_baseline(str(tmp_path), "g", "lib", [object()], 44, 64, n_seeds=10)
    assert sb["cores"] == 32 and sb["seconds"] == 1.0 and set(sb["seconds_by_threads"]) == {12, 32}
    assert list(sb["crashed_by_threads"]) == [16] and "signal 11" in sb["crashed_by_threads"][16]
    assert sb["expansions"] == 150 + 2 * 10 and sb["value"] == 170.0
