#!/usr/bin/env python3
"""reads.fa -> contigs through megagta.py at the north star's own size (100 M x 150 bp, rplB + nirK, k = 30,36,45) on one MI355X: ours only
(the reference needs about two hours for this input: 356.7 s at 5 M reads, profiles/r03/e2e_5M_same_sample.json).  The reads file is
written chunk by chunk (4 M reads at a time), so the host never holds it.  python scripts/e2e_full_size.py [n_reads] [out_dir_for_logs]"""
import json, os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from megagta_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
logdir = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out")
os.makedirs(logdir, exist_ok=True)
d = tempfile.mkdtemp(prefix="mgta_e2e_full_")
need = n * (163 + 40 + 3 * 80) * 1.1                                   # reads.fa + library + three graphs' files + contigs, bytes (generous)
free = shutil.disk_usage(d).free
print(f"scratch {d}: {free / 1e9:.0f} GB free, about {need / 1e9:.0f} GB needed", flush=True)
if free < need:
    sys.exit("not enough scratch space for this size")
try:
    t = time.time()
    fa = open(d + "/reads.fa", "wb")
    L, width = 150, 9
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)

    def sink(first, codes):
        m = codes.shape[0]
        rec = np.empty((m, 2 + width + 1 + L + 1), dtype=np.uint8)
        rec[:, 0], rec[:, 1] = ord(">"), ord("r")
        ids = np.arange(first, first + m, dtype=np.int64)
        for dg in range(width):
            rec[:, 2 + width - 1 - dg] = (ids % 10 + ord("0")).astype(np.uint8)
            ids //= 10
        rec[:, 2 + width] = ord("\n")
        rec[:, 3 + width:3 + width + L] = lut[codes]
        rec[:, -1] = ord("\n")
        rec.tofile(fa)
        if first % 20_000_000 == 0:
            print(f"  reads written: {first + m}", flush=True)

    mg = synth.make_metagenome_device(n, L, (("rplB", 277), ("nirK", 360)), seed=1000 + n % 997, device="cuda:0", host_sample=0, on_chunk=sink)
    fa.close()
    gl = synth.write_gene_models(mg.genes, d + "/models")
    del mg
    import torch
    torch.cuda.empty_cache()
    print(f"reads.fa: {os.path.getsize(d + '/reads.fa') / 1e9:.1f} GB in {time.time() - t:.0f} s", flush=True)
    t = time.time()
    import threading
    done = threading.Event()

    def janitor():
        """the box's scratch disk holds 84 GB and a 100 M-read run writes more (reads.fa 16 GB, the library 4 GB, three graphs of 12-13 GB,
        11 GB of intermediate contigs, 10 GB of raw + 10 GB of filtered contigs): what no later step reads is removed as the run passes it --
        reads.fa once the library is built, a k's graph files once the next k is being built (its contigs stay)"""
        log = d + "/out/log"
        gone = set()
        while not done.wait(5):
            try:
                txt = open(log, errors="replace").read()
            except OSError:
                continue
            for mark, paths in (("Building sdbg for k = 29", [d + "/reads.fa"]), ("Building sdbg for k = 35", [d + "/out/k29/29.sdbg"]),
                                ("Building sdbg for k = 44", [d + "/out/k35/35.sdbg"])):
                if mark in txt and mark not in gone:
                    gone.add(mark)
                    freed = 0
                    for pth in paths:
                        for f in ([pth] if os.path.isfile(pth) else [os.path.join(os.path.dirname(pth), x) for x in os.listdir(os.path.dirname(pth)) if x.startswith(os.path.basename(pth) + ".")]):
                            try:
                                freed += os.path.getsize(f); os.remove(f)
                            except OSError:
                                pass
                    print(f"  janitor: '{mark}' reached, {freed / 1e9:.1f} GB of files no later step reads removed", flush=True)
    threading.Thread(target=janitor, daemon=True).start()

    def heartbeat():                                                    # (a line a minute: a silent call is taken to be hung)
        while not done.wait(60):
            print(f"  driver running: {time.time() - t:.0f} s", flush=True)
            try:                                                        # the driver's detailed log so far: a run cut off at the call's limit still leaves it
                shutil.copyfile(d + "/out/log", os.path.join(logdir, f"e2e_{n // 1_000_000}M_out_log.txt"))
            except OSError:
                pass
    threading.Thread(target=heartbeat, daemon=True).start()
    with open(os.path.join(logdir, f"e2e_{n // 1_000_000}M_steps.log"), "w") as lg:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "megagta_amd", "megagta.py"), "-r", d + "/reads.fa", "-g", gl, "-k", "30,36,45", "-o", d + "/out",
                            "-c", "1", "-t", "16"], stdout=subprocess.DEVNULL, stderr=lg, text=True, env={**os.environ, "MGTA_DENOVO_VERBOSE": "1"})
        dt = time.time() - t
        done.set()
        if os.path.exists(d + "/out/log"):
            lg.write("\n==== <out>/log ====\n" + open(d + "/out/log", errors="replace").read())
    if r.returncode != 0:
        sys.exit(f"megagta.py failed ({r.returncode}) after {dt:.0f} s: see the step log")
    if os.environ.get("MEGAGTA_STOP_BEFORE_SEARCH"):
        print(f"reads -> seeds: {dt:.1f} s (stopped before the search)", flush=True)
        sys.exit(0)
    nc = {g: sum(1 for l in open(f"{d}/out/contigs/{g}/nucl_merged.fasta") if l.startswith(">")) for g in ("rplB", "nirK")}
    line = {"reads": n, "read_len": L, "k_list": "30,36,45", "genes": ["rplB", "nirK"], "seconds": dt, "reads_per_s": n / dt, "contigs": nc,
            "note": "megagta.py, one MI355X, default mode (ordered-commit window); reads.fa on the box's scratch disk; the reference was not run at this size"}
    print(json.dumps(line), flush=True)
    with open(os.path.join(logdir, f"e2e_{n // 1_000_000}M_ours.json"), "w") as f:
        json.dump(line, f, indent=1)
finally:
    shutil.rmtree(d, ignore_errors=True)
