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
    open(os.path.join(out, "contigs", g, "nucl_merged.fasta"), "w").write(">c0\nACGT\n>c1\nACGT\n")
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
