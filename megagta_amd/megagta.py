#!/usr/bin/env python3
"""megagta.py — Python-3 driver with the reference driver's CLI, step order, file tree and checkpoints.

Drop-in surface kept (reference src/megagta.py): options `-r/-1/-2/--12 -g -k -c -p -l -m -t -o
--min-contig-len --max-tip-len --no-mercy --mem-flag --keep-tmp-files --continue --verbose
--gpu-mem` (:150-174), `opts.txt`, `tmp/cp.txt` ("<n>\\tdone" per finished step, :380-385), `log`,
`k<K>/<K>.*`, `contigs/<gene>/{nucl,prot}_merged.fasta`, k list decremented by one for the graph
(:815-816), one child process per step with stderr relayed into the log, first non-zero exit aborts.

Every sub-command runs from THIS package's `bin/megagta` (C++ host + libmegagta_hip.so): `buildgraph`, `denovo`, `findstart`, `search`
on the device, `buildlib`, `filterbylen`, `translate` on the host.  `--ref-bin` / $MEGAGTA_REF_BIN (the stock MegaGTA executable) is
kept for steps a future reference version may add; nothing needs it today.
"""
from __future__ import annotations

import getopt
import logging
import multiprocessing
import os
import subprocess
import sys
import time
from datetime import datetime

VERSION = "MegaGTA (megagta_amd host driver) v0.1"
USAGE = """Usage:
  megagta.py [options] {-1 <pe1> -2 <pe2> | --12 <pe12> | -r <se>} -g <gene_list.txt> [-o <out_dir>]
    -k/--k-list 30,36,45   -c/--min-count 1   -p/--prune-len 20   -l/--low-cov-penalty 0.5
    -m/--memory 0.9        -t/--num-cpu-threads N   --min-contig-len 450   --max-tip-len 150
    --no-mercy  --mem-flag 1  --gpu-mem BYTES  --keep-tmp-files  --continue  --verbose
    --ref-bin PATH   stock `megagta` binary for the steps outside the accelerated path"""


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
        self.ref_bin = os.environ.get("MEGAGTA_REF_BIN", "")


opt = Opt()
cp = 0

LONG = ["help", "read=", "12=", "out-dir=", "memory=", "gpu-mem=", "min-contig-len=", "num-cpu-threads=", "kmin-1pass", "k-list=",
        "min-count=", "max-tip-len=", "no-mercy", "keep-tmp-files", "mem-flag=", "version", "verbose", "continue", "gene-list=",
        "prune-len=", "low-cov-penalty=", "ref-bin=", "bin="]


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
        elif o == "--ref-bin": opt.ref_bin = v
        elif o == "--bin": opt.bin = v
        else:
            raise Usage("Invalid option " + o)
    opt.temp_dir = opt.out_dir + "tmp/"
    if need_continue:
        prepare_continue()
    elif not opt.continue_mode and os.path.exists(opt.out_dir):
        raise Usage("Output directory " + opt.out_dir + " already exists, please change the parameter -o to another value to avoid overwriting.")


def prepare_continue():
    """re-read opts.txt and the last finished checkpoint (reference :321-351)"""
    opt.continue_mode = True
    if not os.path.exists(opt.out_dir + "opts.txt"):
        raise Usage("Cannot find " + opt.out_dir + "opts.txt, nothing to continue")
    with open(opt.out_dir + "opts.txt") as f:
        argv = [l.rstrip("\n") for l in f if l.strip()]
    parse_opt(argv)
    opt.last_cp = -1
    if os.path.exists(opt.temp_dir + "cp.txt"):
        with open(opt.temp_dir + "cp.txt") as f:
            for line in f:
                a = line.split()
                if len(a) == 2 and a[1] == "done":
                    opt.last_cp = int(a[0])


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


def write_cp():
    global cp
    with open(opt.temp_dir + "cp.txt", "a") as f:
        f.write(f"{cp}\tdone\n")
    cp += 1


def should_run():
    return (not opt.continue_mode) or cp > opt.last_cp


def run_step(cmd, what, stdin=None, stdout=None):
    """one child process per step, stderr relayed line by line into the log (reference :563-576)"""
    logging.info("--- [%s] %s ---" % (datetime.now().strftime("%c"), what))
    logging.debug("cmd: " + " ".join(cmd))
    try:
        p = subprocess.Popen(cmd, stdin=stdin, stdout=stdout, stderr=subprocess.PIPE)
    except OSError:
        logging.error("Error: sub-program %s not found" % cmd[0])
        sys.exit(1)
    for line in p.stderr:
        logging.debug(line.decode(errors="replace").rstrip())
    ret = p.wait()
    if ret != 0:
        logging.error("Error occurs when running \"%s\", please refer to %s for detail" % (what, log_file()))
        logging.error("[Exit code %d]" % ret)
        sys.exit(ret)


def need_ref(step):
    if not opt.ref_bin or not os.path.exists(opt.ref_bin):
        logging.error("step '%s' is outside the accelerated path: give the stock megagta binary with --ref-bin / $MEGAGTA_REF_BIN" % step)
        sys.exit(1)
    return opt.ref_bin


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
        run_step(cmd, "Building sdbg for k = %d" % k)
    write_cp()


def assemble(k):
    if should_run():
        nxt = opt.k_list[opt.k_list.index(k) + 1]
        run_step([opt.bin, "denovo", "-s", graph_prefix(k), "-o", graph_prefix(k), "-t", str(opt.num_cpu_threads),
                  "--min_standalone", "400", "--max_tip_len", str(opt.max_tip_len), "--min_contig", str(nxt + 1)],
                 "De novo assembling contigs from SdBG for k = %d" % k, stdout=subprocess.PIPE)
    write_cp()


def find_seed(k, gene):
    if should_run():
        par = [opt.gene_info[gene][2], opt.lib + ".bin", str(k + 1), str(opt.num_cpu_threads)]
        i = opt.k_list.index(k)
        if i > 0:
            par.append(contig_file(opt.k_list[i - 1]))
        with open(graph_prefix(k) + "_" + gene + "_starting_kmers.txt", "w") as out:
            run_step([opt.bin, "findstart"] + par, "Finding starting kmers for %s k = %d" % (gene, k), stdout=out)
    write_cp()


def search_contigs(k):
    run_it = should_run()
    if run_it:
        run_step([opt.bin, "search", graph_prefix(k), opt.gene_list, graph_prefix(k), graph_prefix(k), str(opt.prune_len),
                  str(opt.low_cov_penalty), str(min(12, opt.num_cpu_threads))], "Searching contigs for k = %d" % k)
    write_cp()
    os.makedirs(opt.out_dir + "contigs", exist_ok=True)
    for gene in opt.gene_info:
        d = opt.out_dir + "contigs/" + gene
        os.makedirs(d, exist_ok=True)
        if should_run():
            with open(graph_prefix(k) + "_raw_contigs_" + gene + ".fasta") as fin, open(d + "/nucl_merged.fasta", "w") as fout:
                run_step([opt.bin, "filterbylen", str(opt.min_contig_len)],
                         "Filtering contigs with minimum length = %d" % opt.min_contig_len, stdin=fin, stdout=fout)
        write_cp()
        if should_run():
            with open(d + "/prot_merged.fasta", "w") as fout:
                run_step([opt.bin, "translate", d + "/nucl_merged.fasta"], "Translating nucl contigs to aa contigs", stdout=fout)
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
                search_contigs(k)
        logging.info("--- [%s] ALL DONE. Time elapsed: %f seconds ---" % (datetime.now().strftime("%c"), time.time() - t0))
        return 0
    except Usage as e:
        print("megagta.py: " + str(e), file=sys.stderr)
        return 2


if __name__ == "__main__":
    sys.exit(main())
