#!/usr/bin/env python3
"""megagta.py — Python-3 driver with the reference driver's CLI, step order, file tree and checkpoints.

Drop-in surface kept (reference src/megagta.py): options `-r/-1/-2/--12 -g -k -c -p -l -m -t -o
--min-contig-len --max-tip-len --no-mercy --mem-flag --keep-tmp-files --continue --verbose
--gpu-mem` (:150-174), `opts.txt`, `tmp/cp.txt` ("<n>\\tdone" per finished step, :380-385), `log`,
`k<K>/<K>.*`, `contigs/<gene>/{nucl,prot}_merged.fasta`, k list decremented by one for the graph
(:815-816), one child process per step with stderr relayed into the log, first non-zero exit aborts.

Every sub-command runs from THIS package's `bin/megagta` (C++ host + libmegagta_hip.so): `buildgraph`, `denovo`, `findstart`, `search`
on the device, `buildlib`, `filterbylen`, `translate` on the host.  By default the steps are requests to ONE worker process (`megagta
serve`): same sub-commands, same files and checkpoints, but the device context, the unpacked read library and the graph of the last
`buildgraph` stay where they are between steps.  `--one-process-per-step` (or a `--bin` without `serve`, e.g. the stock MegaGTA
executable) runs one child process per step like the reference driver.
"""
from __future__ import annotations

import getopt
import logging
import multiprocessing
import os
import subprocess
import sys
import threading
import time
from datetime import datetime

VERSION = "MegaGTA (megagta_amd host driver) v0.1"
USAGE = """Usage:
  megagta.py [options] {-1 <pe1> -2 <pe2> | --12 <pe12> | -r <se>} -g <gene_list.txt> [-o <out_dir>]
    -k/--k-list 30,36,45   -c/--min-count 1   -p/--prune-len 20   -l/--low-cov-penalty 0.5
    -m/--memory 0.9        -t/--num-cpu-threads N   --min-contig-len 450   --max-tip-len 150
    --no-mercy  --mem-flag 1  --gpu-mem BYTES  --keep-tmp-files  --continue  --verbose
    --bin PATH       the multi-call `megagta` executable (default: this package's bin/megagta)
    --gpus N         GPUs of this node.  `search`: one process per GPU (torch.distributed over RCCL), seeds sharded by gene first, one
                     all-gather of contigs (megagta_amd/search_dist.py).  `buildgraph`: one process per GPU, each builds its share of the
                     65536 prefix buckets and writes it as <prefix>.sdbg.<rank> (no exchange at all).  denovo / findstart run on GPU 0
    --one-process-per-step   start every step as its own process, as the reference driver does (default: one worker process,
                     `megagta serve`, runs all steps and keeps the device context, the read library and the last graph between them)"""


class Usage(Exception):
    pass


class Opt:
    def __init__(self):
        self.host_mem = 0.9
        self.gpu_mem = 0
        self.out_dir = "./megagta_out/"
        self.min_contig_len = 450
        self.max_tip_len = 150
        self.prune_len = 20
        self.low_cov_penalty = 0.5
        self.k_list = [30, 36, 45]
        self.min_count = 1
        self.no_mercy = False
        self.num_cpu_threads = 0
        self.keep_tmp_files = False
        self.mem_flag = 1
        self.continue_mode = False
        self.last_cp = -1
        self.verbose = False
        self.pe1, self.pe2, self.pe12, self.se = [], [], [], []
        self.gene_list = ""
        self.gene_info = {}
        self.bin = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "megagta")
        self.one_process_per_step = False
        self.gpus = 1


opt = Opt()
cp = 0

LONG = ["help", "read=", "12=", "out-dir=", "memory=", "gpu-mem=", "min-contig-len=", "num-cpu-threads=", "kmin-1pass", "k-list=",
        "min-count=", "max-tip-len=", "no-mercy", "keep-tmp-files", "mem-flag=", "version", "verbose", "continue", "gene-list=",
        "prune-len=", "low-cov-penalty=", "bin=", "one-process-per-step", "gpus="]


def parse_opt(argv):
    try:
        opts, _ = getopt.getopt(argv, "hm:o:r:t:v1:2:l:k:c:g:p:", LONG)
    except getopt.GetoptError as e:
        raise Usage(VERSION + "\n" + str(e))
    if not opts:
        raise Usage(VERSION + "\n" + USAGE)
    need_continue = False
    for o, v in opts:
        if o in ("-h", "--help"):
            print(VERSION + "\n" + USAGE)
            sys.exit(0)
        elif o in ("-o", "--out-dir"):
            if not opt.continue_mode:
                opt.out_dir = v + "/"
        elif o in ("-m", "--memory"): opt.host_mem = float(v)
        elif o == "--gpu-mem": opt.gpu_mem = int(float(v))
        elif o == "--min-contig-len": opt.min_contig_len = int(v)
        elif o in ("-t", "--num-cpu-threads"): opt.num_cpu_threads = int(v)
        elif o == "--kmin-1pass": pass
        elif o in ("-k", "--k-list"): opt.k_list = sorted(int(x) for x in v.split(","))
        elif o in ("-c", "--min-count"): opt.min_count = int(v)
        elif o == "--max-tip-len": opt.max_tip_len = int(v)
        elif o == "--no-mercy": opt.no_mercy = True
        elif o == "--keep-tmp-files": opt.keep_tmp_files = True
        elif o == "--mem-flag": opt.mem_flag = int(v)
        elif o in ("-v", "--version"):
            print(VERSION)
            sys.exit(0)
        elif o == "--verbose": opt.verbose = True
        elif o == "--continue":
            if not opt.continue_mode:
                need_continue = True
        elif o in ("-r", "--read"): opt.se += v.split(",")
        elif o == "-1": opt.pe1 += v.split(",")
        elif o == "-2": opt.pe2 += v.split(",")
        elif o == "--12": opt.pe12 += v.split(",")
        elif o in ("-g", "--gene-list"): opt.gene_list = v
        elif o in ("-p", "--prune-len"): opt.prune_len = int(v)
        elif o in ("-l", "--low-cov-penalty"): opt.low_cov_penalty = float(v)
        elif o == "--bin": opt.bin = v
        elif o == "--one-process-per-step": opt.one_process_per_step = True
        elif o == "--gpus": opt.gpus = int(v)
        else:
            raise Usage("Invalid option " + o)
    opt.temp_dir = opt.out_dir + "tmp/"
    if need_continue:
        prepare_continue()
    elif not opt.continue_mode and os.path.exists(opt.out_dir):
        raise Usage("Output directory " + opt.out_dir + " already exists, please change the parameter -o to another value to avoid overwriting.")


def prepare_continue():
    """re-read opts.txt and the last finished checkpoint (reference :321-351): every option but -o comes from opts.txt, parsed into a
    fresh option set (options given next to --continue are ignored, as the reference says it does); without an opts.txt the run goes on
    in normal mode"""
    global opt
    if not os.path.exists(opt.out_dir + "opts.txt"):
        print("Cannot find " + opt.out_dir + "opts.txt", file=sys.stderr)
        print("Please check whether the output directory is correctly set by \"-o\"", file=sys.stderr)
        print("Now switching to normal mode.", file=sys.stderr)
        return
    print("Continue mode activated. Ignore all options other than -o/--out-dir.", file=sys.stderr)
    with open(opt.out_dir + "opts.txt") as f:
        argv = [l.rstrip("\n") for l in f if l.strip()]
    out_dir = opt.out_dir
    opt = Opt()
    opt.out_dir = out_dir
    opt.continue_mode = True
    parse_opt(argv)
    opt.last_cp = -1
    if os.path.exists(opt.temp_dir + "cp.txt"):
        with open(opt.temp_dir + "cp.txt") as f:
            for line in f:
                a = line.split()
                if len(a) == 2 and a[1] == "done":
                    opt.last_cp = int(a[0])
    print("Continue from check point " + str(opt.last_cp), file=sys.stderr)


def detect_available_mem():
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemTotal"):
                    return int(line.split()[1]) * 1024
    except OSError:
        pass
    return 0


def check_opt():
    if opt.host_mem <= 0:
        raise Usage("Please specify a positive number for -m flag.")
    if opt.host_mem < 1:
        total = detect_available_mem()
        if total <= 0:
            raise Usage("Failed to detect available memory. Please specify the value in bytes using -m flag.")
        opt.host_mem = int(total * opt.host_mem)
    else:
        opt.host_mem = int(opt.host_mem)
    if not opt.k_list:
        raise Usage("k list should not be empty!")
    if opt.k_list[0] < 15 or opt.k_list[-1] > 127:
        raise Usage("All k's should be in range [15, 127]")
    if opt.k_list[-1] % 3 != 0:
        raise Usage("The last k must be a multiple of 3")
    if opt.min_count <= 0:
        raise Usage("min_count must be greater than 0.")
    if opt.min_count == 1:
        opt.no_mercy = True
    ncpu = multiprocessing.cpu_count()
    if opt.num_cpu_threads > ncpu or opt.num_cpu_threads == 0:
        opt.num_cpu_threads = ncpu
    opt.num_cpu_threads = max(2, opt.num_cpu_threads)
    if opt.gene_list == "":
        raise Usage("--gene-list could not be empty")
    if opt.prune_len <= 0:
        raise Usage("prune length should be >= 1")
    if not 0 <= opt.low_cov_penalty <= 1:
        raise Usage("low coverage penalty should be between [0, 1]")
    if len(opt.pe1) != len(opt.pe2):
        raise Usage("Number of paired-end files not match!")
    for r in opt.pe1 + opt.pe2 + opt.se + opt.pe12:
        if not os.path.exists(r):
            raise Usage("Cannot find file " + r)
    if not (opt.pe1 or opt.se or opt.pe12):
        raise Usage("No input files or input command!")
    if not os.path.exists(opt.bin):
        raise Usage("Cannot find sub-program " + opt.bin + " (build it: make -C megagta_amd/csrc)")


def graph_prefix(k): return f"{opt.out_dir}k{k}/{k}"
def contig_file(k): return graph_prefix(k) + ".contigs.fa"
def log_file(): return opt.out_dir + "log"


deferred_cp = []      # checkpoints of steps whose files a worker thread is still writing (`buildgraph` in the worker)


def flush_deferred_cp():
    """the worker's background writer has finished (request "sync"): only now does cp.txt say that those graphs are built.  The reference
    writes a checkpoint after the step's files are complete (megagta.py:380-385,586); a run killed in between re-builds the graph."""
    global deferred_cp
    if not deferred_cp:
        return
    if worker is not None:
        ret = worker.request(["sync"])
        if ret != 0:
            fail_step("writing the graph / contig files", ret)
    with open(opt.temp_dir + "cp.txt", "a") as f:
        for line in deferred_cp:
            f.write(line)
    deferred_cp = []


def write_cp(defer=False):
    global cp
    line = f"{cp}\tdone\n"
    cp += 1
    if defer and worker is not None:
        deferred_cp.append(line)
        return
    flush_deferred_cp()
    with open(opt.temp_dir + "cp.txt", "a") as f:
        f.write(line)


def should_run():
    return (not opt.continue_mode) or cp > opt.last_cp


class Worker:
    """`megagta serve`: requests = one tab-separated argv per line (+ "<path" / ">path" redirections), reply "DONE <rc>"."""

    def __init__(self, binary):
        self.p = subprocess.Popen([binary, "serve"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        self.t = threading.Thread(target=self._relay, daemon=True)
        self.t.start()

    def _relay(self):
        for line in self.p.stderr:
            logging.debug(line.decode(errors="replace").rstrip())

    def request(self, fields):
        try:
            self.p.stdin.write(("\t".join(fields) + "\n").encode())
            self.p.stdin.flush()
            while True:                          # anything that is not the reply (there should be nothing: the worker keeps its steps'
                raw = self.p.stdout.readline()   # stdout off this pipe) is logged and skipped; an empty read = the worker is gone
                reply = raw.decode(errors="replace").split()
                if not raw or (len(reply) == 2 and reply[0] == "DONE"):
                    break
                logging.debug("worker: " + raw.decode(errors="replace").rstrip())
        except (BrokenPipeError, OSError):
            reply = []
        if len(reply) == 2 and reply[0] == "DONE":
            return int(reply[1])
        try:                                     # the worker died in the step (a fatal error prints its reason and exits)
            ret = self.p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            self.p.kill()
            ret = self.p.wait()
        return ret if ret != 0 else 1

    def close(self):
        try:
            self.p.stdin.write(b"quit\n")
            self.p.stdin.flush()
            self.p.stdin.close()
        except (BrokenPipeError, OSError):
            pass
        self.p.wait()
        self.t.join(timeout=5)


worker = None


def start_worker():
    """one worker for all steps when the executable has `serve` (ours has; the stock MegaGTA binary has not)"""
    global worker
    if opt.one_process_per_step:
        return
    try:
        w = Worker(opt.bin)
    except OSError:
        return
    if w.request(["dumpversion", ">" + os.devnull]) == 0:
        worker = w
    else:
        logging.debug("%s has no `serve`: one process per step" % opt.bin)


def run_step(cmd, what, stdin_path=None, stdout_path=None):
    """one step: a request to the worker, or one child process (reference :563-576); stderr is relayed line by line into the log"""
    logging.info("--- [%s] %s ---" % (datetime.now().strftime("%c"), what))
    logging.debug("cmd: " + " ".join(cmd))
    if worker is not None:
        fields = cmd[1:] + (["<" + stdin_path] if stdin_path else []) + ([">" + stdout_path] if stdout_path else [])
        ret = worker.request(fields)
    else:
        fin = open(stdin_path, "rb") if stdin_path else None
        fout = open(stdout_path, "wb") if stdout_path else None
        try:
            p = subprocess.Popen(cmd, stdin=fin, stdout=fout, stderr=subprocess.PIPE)
        except OSError:
            logging.error("Error: sub-program %s not found" % cmd[0])
            sys.exit(1)
        for line in p.stderr:
            logging.debug(line.decode(errors="replace").rstrip())
        ret = p.wait()
        for f in (fin, fout):
            if f:
                f.close()
    if ret != 0:
        logging.error("Error occurs when running \"%s\", please refer to %s for detail" % (what, log_file()))
        logging.error("[Exit code %d]" % ret)
        if worker is not None:
            worker.close()
        sys.exit(ret)


def build_lib():
    """tmp/reads.lib (2 lines per library, :395-434) then `buildlib` -> reads.lib.bin/.lib_info"""
    opt.lib = opt.temp_dir + "reads.lib"
    if should_run():
        with open(opt.lib, "w") as f:
            for i in range(len(opt.pe12)):
                f.write(opt.pe12[i] + "\ninterleaved " + os.path.abspath(opt.pe12[i]) + "\n")
            for i in range(len(opt.pe1)):
                f.write(opt.pe1[i] + "," + opt.pe2[i] + "\npe " + os.path.abspath(opt.pe1[i]) + " " + os.path.abspath(opt.pe2[i]) + "\n")
            for r in opt.se:
                f.write(r + "\nse " + os.path.abspath(r) + "\n")
        run_step([opt.bin, "buildlib", opt.lib, opt.lib], "Converting reads to binaries")
    write_cp()


def parse_gene_list():
    with open(opt.gene_list) as f:
        for line in f:
            a = line.split()
            if len(a) >= 4:
                opt.gene_info[a[0]] = (a[1], a[2], a[3])


def build_graph(k, assist):
    if should_run():
        os.makedirs(f"{opt.out_dir}k{k}", exist_ok=True)
        cmd = [opt.bin, "buildgraph", "-k", str(k), "-m", str(opt.min_count), "--host_mem", str(opt.host_mem), "--mem_flag",
               str(opt.mem_flag), "--gpu_mem", str(opt.gpu_mem), "--output_prefix", graph_prefix(k), "--num_cpu_threads",
               str(opt.num_cpu_threads), "--num_output_threads", str(max(1, opt.num_cpu_threads // 3)), "--read_lib_file", opt.lib]
        if not opt.no_mercy:
            cmd.append("--need_mercy")
        if assist:
            cmd += ["--assist_seq", assist]
        if opt.gpus > 1:
            run_multi_gpu_build(cmd, k)
            write_cp()
        else:
            run_step(cmd, "Building sdbg for k = %d" % k)
            write_cp(defer=True)      # (the worker replies while a thread still writes PREFIX.sdbg.*: the checkpoint waits for the files)
        return
    write_cp()


def fail_step(what, ret):
    logging.error("Error occurs when %s, please refer to %s for detail" % (what, log_file()))
    logging.error("[Exit code %d]" % ret)
    if worker is not None:
        worker.close()
    sys.exit(ret)


def release_worker_memory():
    """the worker hands its device memory back before other processes use GPU 0; a worker that cannot answer is dropped"""
    global worker
    flush_deferred_cp()                                  # (a writer failure is the build's failure, not a refusal to release)
    if worker is not None and worker.request(["release"]) != 0:
        logging.debug("the worker did not release its memory: it is stopped, the remaining steps run one process each")
        worker.close()
        worker = None


def run_multi_gpu_build(cmd, k):
    """`buildgraph` over opt.gpus GPUs: the same command line once per GPU (MEGAGTA_RANK / MEGAGTA_WORLD / MEGAGTA_DEVICE), every rank
    builds its share of the prefix buckets into <prefix>.sdbg.<rank>; `sdbgmerge` then writes the index that names the files"""
    what = "Building sdbg for k = %d on %d GPUs" % (k, opt.gpus)
    logging.info("--- [%s] %s ---" % (datetime.now().strftime("%c"), what))
    release_worker_memory()
    logging.debug("cmd (x%d ranks): %s" % (opt.gpus, " ".join(cmd)))
    procs = []
    for r in range(opt.gpus):
        # rank r on GPU r (MEGAGTA_DEVICE in the environment: every rank on that device -- tests on a one-GPU box)
        env = dict(os.environ, MEGAGTA_RANK=str(r), MEGAGTA_WORLD=str(opt.gpus), MEGAGTA_DEVICE=os.environ.get("MEGAGTA_DEVICE", str(r)))
        procs.append(subprocess.Popen(cmd, env=env, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL))
    relays = [threading.Thread(target=lambda p=p: [logging.debug(l.decode(errors="replace").rstrip()) for l in p.stderr], daemon=True) for p in procs]
    for t in relays:
        t.start()
    rets = [p.wait() for p in procs]
    for t in relays:
        t.join(timeout=5)
    bad = [r for r in rets if r != 0]
    if bad:
        fail_step("running %s" % what, bad[0])
    ret = subprocess.call([opt.bin, "sdbgmerge", graph_prefix(k), str(opt.gpus)])
    if ret != 0:
        fail_step("merging the index of the %d graph files" % opt.gpus, ret)


def assemble(k):
    if should_run():
        nxt = opt.k_list[opt.k_list.index(k) + 1]
        run_step([opt.bin, "denovo", "-s", graph_prefix(k), "-o", graph_prefix(k), "-t", str(opt.num_cpu_threads),
                  "--min_standalone", "400", "--max_tip_len", str(opt.max_tip_len), "--min_contig", str(nxt + 1)],
                 "De novo assembling contigs from SdBG for k = %d" % k, stdout_path=os.devnull)
        write_cp(defer=True)          # (the worker replies while a thread still writes PREFIX.contigs.fa: the checkpoint waits for the file)
        return
    write_cp()


def find_seed(k, gene):
    if should_run():
        par = [opt.gene_info[gene][2], opt.lib + ".bin", str(k + 1), str(opt.num_cpu_threads)]
        i = opt.k_list.index(k)
        if i > 0:
            par.append(contig_file(opt.k_list[i - 1]))
        run_step([opt.bin, "findstart"] + par, "Finding starting kmers for %s k = %d" % (gene, k),
                 stdout_path=graph_prefix(k) + "_" + gene + "_starting_kmers.txt")
        # (its own file is complete, but cp.txt is an ordered log: while the graph / contig files of the steps before are still on their way
        # to the disk this line waits with theirs -- 100 M reads: the "sync" it used to trigger stood 8 s in front of the search)
        write_cp(defer=True)
        return
    write_cp()


def run_multi_gpu_search(par, k):
    """the search step on opt.gpus GPUs: its own processes (one per GPU), same arguments and files as `megagta search`"""
    logging.info("--- [%s] Searching contigs for k = %d on %d GPUs ---" % (datetime.now().strftime("%c"), k, opt.gpus))
    release_worker_memory()                              # the worker gives its device memory back while the ranks run
    # torchrun picks the rendezvous port itself (--standalone = c10d on a free port of this host): nothing to race for
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node", str(opt.gpus),
           os.path.join(os.path.dirname(os.path.abspath(__file__)), "search_dist.py")] + par
    logging.debug("cmd: " + " ".join(cmd))
    p = subprocess.Popen(cmd, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL)
    for line in p.stderr:
        logging.debug(line.decode(errors="replace").rstrip())
    ret = p.wait()
    if ret != 0:
        fail_step("searching contigs for k = %d on %d GPUs" % (k, opt.gpus), ret)


def filter_and_translate_side_by_side(k):
    """the two host-only text filters of every gene (filter_by_len.cpp, translate.cpp): the genes' chains are independent, so they run as
    child processes side by side (100 M reads: 8 + 12 s one after the other through the worker); the log lines and the checkpoints keep the
    reference's order (:705-710), written once the chains have ended"""
    results = {}

    def chain(gene):
        d = opt.out_dir + "contigs/" + gene
        lines = []
        results[gene] = lines                            # (set first: whatever happens below, the gene has an entry)
        for cmd, fin_path, fout_path in (([opt.bin, "filterbylen", str(opt.min_contig_len)], graph_prefix(k) + "_raw_contigs_" + gene + ".fasta", d + "/nucl_merged.fasta"),
                                         ([opt.bin, "translate", d + "/nucl_merged.fasta"], None, d + "/prot_merged.fasta")):
            # an exception in this thread (the raw contigs missing, the output not writable, the binary not startable) is a failed
            # step like a non-zero exit code: it must not leave the gene without its two result lines (advisor r5)
            try:
                os.makedirs(d, exist_ok=True)
                with open(fout_path, "wb") as fout:
                    fin = open(fin_path, "rb") if fin_path else None
                    try:
                        p = subprocess.run(cmd, stdin=fin, stdout=fout, stderr=subprocess.PIPE)
                    finally:
                        if fin:
                            fin.close()
                lines.append((" ".join(cmd), p.stderr.decode(errors="replace"), p.returncode))
            except Exception as e:                       # noqa: BLE001 -- reported as the step's failure by the main thread
                lines.append((" ".join(cmd), "%s: %s" % (type(e).__name__, e), 1))
            if lines[-1][2] != 0:
                break

    threads = [threading.Thread(target=chain, args=(gene,)) for gene in opt.gene_info]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    steps = ("Filtering contigs with minimum length = %d" % opt.min_contig_len, "Translating nucl contigs to aa contigs")
    for gene in opt.gene_info:
        lines = results.get(gene, [])
        for what, (cmd, err, ret) in zip(steps, lines):
            logging.info("--- [%s] %s ---" % (datetime.now().strftime("%c"), what))
            logging.debug("cmd: " + cmd)
            for line in err.splitlines():
                (logging.debug if ret == 0 else logging.error)(line.rstrip())
            if ret != 0:
                fail_step("running \"%s\" for %s" % (what, gene), ret)
            write_cp()
        if len(lines) != len(steps):                     # exactly two result lines per gene, or the run stops here (never a silent exit 0
            fail_step("running \"%s\" for %s (no result)" % (steps[len(lines)], gene), 1)   # with the gene's files and checkpoints missing)


def search_contigs(k):
    """search, then per gene filterbylen + translate.  Checkpoints as the reference writes them (:680-760): the two filters of every
    gene have their own, written INSIDE the search step, the search's own comes last -- a finished run continues identically under
    either driver"""
    if should_run():
        par = [graph_prefix(k), opt.gene_list, graph_prefix(k), graph_prefix(k), str(opt.prune_len), str(opt.low_cov_penalty),
               str(min(12, opt.num_cpu_threads))]
        if opt.gpus > 1:
            run_multi_gpu_search(par, k)
        else:
            run_step([opt.bin, "search"] + par, "Searching contigs for k = %d" % k)
        os.makedirs(opt.out_dir + "contigs", exist_ok=True)
        if len(opt.gene_info) > 1 and not opt.continue_mode and worker is not None:
            filter_and_translate_side_by_side(k)
            write_cp()
            return
        for gene in opt.gene_info:
            d = opt.out_dir + "contigs/" + gene
            os.makedirs(d, exist_ok=True)
            if should_run():
                run_step([opt.bin, "filterbylen", str(opt.min_contig_len)], "Filtering contigs with minimum length = %d" % opt.min_contig_len,
                         stdin_path=graph_prefix(k) + "_raw_contigs_" + gene + ".fasta", stdout_path=d + "/nucl_merged.fasta")
            write_cp()
            if should_run():
                run_step([opt.bin, "translate", d + "/nucl_merged.fasta"], "Translating nucl contigs to aa contigs",
                         stdout_path=d + "/prot_merged.fasta")
            write_cp()
    write_cp()


def main(argv=None):
    argv = sys.argv if argv is None else argv
    try:
        t0 = time.time()
        parse_opt(argv[1:])
        check_opt()
        os.makedirs(opt.out_dir, exist_ok=True)
        os.makedirs(opt.temp_dir, exist_ok=True)
        logging.basicConfig(level=logging.NOTSET, format="%(message)s", filename=log_file(), filemode="a")
        console = logging.StreamHandler()
        console.setLevel(logging.NOTSET if opt.verbose else logging.INFO)
        console.setFormatter(logging.Formatter("%(message)s"))
        logging.getLogger("").addHandler(console)
        logging.info(VERSION)
        logging.info("--- [%s] Start. Number of CPU threads %d ---" % (datetime.now().strftime("%c"), opt.num_cpu_threads))
        logging.info("--- [%s] k list: %s ---" % (datetime.now().strftime("%c"), ",".join(map(str, opt.k_list))))
        if not opt.continue_mode:
            with open(opt.out_dir + "opts.txt", "w") as f:
                f.write("\n".join(argv[1:]) + "\n")
        start_worker()
        build_lib()
        parse_gene_list()
        opt.k_list = [k - 1 for k in opt.k_list]                      # graph k = CLI k - 1
        for i, k in enumerate(opt.k_list):
            build_graph(k, contig_file(opt.k_list[i - 1]) if i > 0 else "")
            if i != len(opt.k_list) - 1:
                assemble(k)
            else:
                for gene in opt.gene_info:
                    find_seed(k, gene)
                if os.environ.get("MEGAGTA_STOP_BEFORE_SEARCH"):         # (measurements of reads -> seeds at sizes whose search takes minutes)
                    flush_deferred_cp()
                    logging.info("--- [%s] stopped before the search (MEGAGTA_STOP_BEFORE_SEARCH). Time elapsed: %f seconds ---"
                                 % (datetime.now().strftime("%c"), time.time() - t0))
                    if worker is not None:
                        worker.close()
                    return 0
                search_contigs(k)
        flush_deferred_cp()
        if worker is not None:
            worker.close()
        logging.info("--- [%s] ALL DONE. Time elapsed: %f seconds ---" % (datetime.now().strftime("%c"), time.time() - t0))
        return 0
    except Usage as e:
        print("megagta.py: " + str(e), file=sys.stderr)
        return 2


if __name__ == "__main__":
    sys.exit(main())
