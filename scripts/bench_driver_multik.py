"""the whole tool through the process boundary: `megagta.py -k 30,36,45` on synthetic reads, every step from bin/megagta
(buildlib, 3x buildgraph, 2x denovo, 3x findstart, 3x search, filterbylen, translate); prints the wall time of every step from the log.
python scripts/bench_driver_multik.py [n_reads] [min_count]"""
import os, re, subprocess, sys, tempfile, time
sys.path.insert(0, ".")
from megagta_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
mc = sys.argv[2] if len(sys.argv) > 2 else "1"
w = tempfile.mkdtemp(dir="/tmp")
t0 = time.time()
mg = synth.make_metagenome(n, 150, (("rplB", 277),), seed=1)
gl = synth.write_gene_models(mg.genes, os.path.join(w, "genes"))
synth.write_fasta(mg.reads, os.path.join(w, "reads.fa"))
print(f"{n} reads written in {time.time() - t0:.1f} s", flush=True)
t0 = time.time()
r = subprocess.run([sys.executable, os.path.join(ROOT, "megagta_amd", "megagta.py"), "-r", os.path.join(w, "reads.fa"), "-g", gl, "-k", "30,36,45", "-c", mc,
                    "-o", os.path.join(w, "out"), "-t", "32"], capture_output=True, text=True)
wall = time.time() - t0
print("exit", r.returncode, f"wall {wall:.1f} s")
log = open(os.path.join(w, "out", "log")).read()
if r.returncode:
    print(log[-3000:])
    sys.exit(1)
stamp = re.compile(r"^--- \[(.*?)\] (.*?) ---", re.M)
ev = [(time.mktime(time.strptime(m.group(1), "%c")), m.group(2)) for m in stamp.finditer(log)]
for (a, what), (b, _) in zip(ev, ev[1:] + [(ev[0][0] + wall, "")]):
    print(f"  {b - a:6.0f} s  {what[:110]}")
for line in log.splitlines():
    if "device build" in line or "Tips removed" in line or "expansions" in line.lower():
        print("   ", line.strip()[:200])
print("contigs:", open(os.path.join(w, "out", "contigs", "rplB", "nucl_merged.fasta")).read().count(">"))
