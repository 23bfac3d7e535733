"""the whole tool through the process boundary: `megagta.py -k 30,36,45` on synthetic reads, every step from bin/megagta
(buildlib, 3x buildgraph, 2x denovo, 3x findstart, 3x search, filterbylen, translate); prints the wall time of every step from the log.
With --ref the same driver is then run with --bin oracle/_ref/megagta (the reference binary built from /root/reference, same argv
contract, all host cores) on the same files, and the reads->contigs wall times are printed side by side.
python scripts/bench_driver_multik.py [n_reads] [min_count] [--ref] [--genes rplB:277,nirK:360] [--ref-timeout S]"""
import json, os, re, subprocess, sys, tempfile, time
sys.path.insert(0, ".")
from megagta_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import argparse
ap = argparse.ArgumentParser()
ap.add_argument("n_reads", nargs="?", type=int, default=2_000_000)
ap.add_argument("min_count", nargs="?", default="1")
ap.add_argument("--ref", action="store_true")
ap.add_argument("--skip-ours", action="store_true", help="reference leg only (runs without a GPU)")
ap.add_argument("--genes", default="rplB:277")
ap.add_argument("--ref-timeout", type=int, default=900)
A = ap.parse_args()
n, mc, gene_arg, ref_timeout = A.n_reads, A.min_count, A.genes, A.ref_timeout
genes = tuple((g.split(":")[0], int(g.split(":")[1])) for g in gene_arg.split(","))
threads = str(os.cpu_count() or 32)
w = tempfile.mkdtemp(dir="/tmp")
t0 = time.time()
mg = synth.make_metagenome(n, 150, genes, seed=1)
gl = synth.write_gene_models(mg.genes, os.path.join(w, "genes"))
synth.write_fasta(mg.reads, os.path.join(w, "reads.fa"))
print(f"{n} reads of {[g[0] for g in genes]} written in {time.time() - t0:.1f} s", flush=True)
stamp = re.compile(r"^--- \[(.*?)\] (.*?) ---", re.M)


def run(tag, extra, timeout=None):
    out = os.path.join(w, "out_" + tag)
    t0 = time.time()
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "megagta_amd", "megagta.py"), "-r", os.path.join(w, "reads.fa"), "-g", gl, "-k", "30,36,45",
                          "-c", mc, "-o", out, "-t", threads] + extra, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
    while p.poll() is None:                                            # a line a minute: a silent run reads as hung on the GPU box
        try:
            p.wait(timeout=60)
        except subprocess.TimeoutExpired:
            print(f"[{tag}] running, {time.time() - t0:.0f} s", flush=True)
            if timeout and time.time() - t0 > timeout:
                os.killpg(p.pid, 15)                                    # the driver and the step it is running
                p.wait()
                print(f"[{tag}] still running after {timeout} s: stopped (lower bound)", flush=True)
                log = open(os.path.join(out, "log")).read() if os.path.exists(os.path.join(out, "log")) else ""
                done = [m.group(2)[:60] for m in stamp.finditer(log)]
                print(f"[{tag}] steps started: {len(done)}; last: {done[-1] if done else '-'}")
                return None, None
    r = p
    wall = time.time() - t0
    print(f"[{tag}] exit {r.returncode} wall {wall:.1f} s", flush=True)
    log = open(os.path.join(out, "log")).read()
    if r.returncode:
        print(log[-3000:])
        return None, None
    ev = [(time.mktime(time.strptime(m.group(1), "%c")), m.group(2)) for m in stamp.finditer(log)]
    for (a, what), (b, _) in zip(ev, ev[1:] + [(ev[0][0] + wall, "")]):
        print(f"  {b - a:6.0f} s  {what[:110]}")
    for line in log.splitlines():
        if "device build" in line or "Tips removed" in line or "expansions" in line.lower():
            print("   ", line.strip()[:200])
    contigs = {g[0]: open(os.path.join(out, "contigs", g[0], "nucl_merged.fasta")).read().count(">") for g in genes}
    print(f"[{tag}] contigs:", contigs, flush=True)
    return wall, contigs


ours, c_ours = (None, None) if A.skip_ours else run("ours", [])
if ours is None and not A.skip_ours:
    sys.exit(1)
line = {"workload": f"{n} x 150bp synthetic reads, genes {gene_arg}, megagta.py -k 30,36,45 -c {mc}, reads.fa -> contigs (files between steps)",
        "ours_wall_s": round(ours, 2) if ours else None, "contigs": c_ours}
if A.ref:
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "megagta")
    ref, c_ref = run("reference", ["--bin", ref_bin], timeout=ref_timeout)
    line.update({"reference_wall_s": round(ref, 2) if ref else f"> {ref_timeout}", "reference_threads": int(threads), "reference_contigs": c_ref,
                 "speedup": None if not ours else round(ref / ours, 2) if ref else f"> {ref_timeout / ours:.1f}"})
print("E2E " + json.dumps(line))
