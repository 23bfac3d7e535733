"""Profile-HMM host logic: HMMER3 text -> the fp64 tables the A* kernels stage in LDS.

Mirrors Parser::readHMM with normalized=true (hmmer3b_parser.h:19-201: natural logs, p = exp(-x),
`*` -> p = 0 -> -inf, match = ln(p/compo_j), insert emissions 0 except -inf at node M, node 0 has no
match line) and MostProbablePath (most_probable_path.h:48-118, insert branch disabled at :100).
`exp`/`log` are libm's, as in the reference.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

NEG_INF = float("-inf")
MM, MI, MD, IM, II, DM, DD = range(7)   # profile_hmm.h:25


@dataclass
class ProfileHMM:
    name: str
    M: int
    A: int
    alpha: np.ndarray      # int32 [127]: residue letter (either case) -> column, -1 = unknown
    compo: np.ndarray      # [A] probabilities
    msc: np.ndarray        # [M+1, A]  (row 0 unused: msc(0, .) = -inf, profile_hmm.h:58-64)
    isc: np.ndarray        # [M+1, A]
    tsc: np.ndarray        # [7, M+1]
    max_match: np.ndarray  # [M+1]
    h: np.ndarray          # [3, M+1]  A* heuristic from state m / i / d


def _p(tok: str) -> float:
    return 0.0 if tok == "*" else math.exp(-1 * float(tok))


def _ln(p: float) -> float:
    return math.log(p) if p > 0.0 else NEG_INF


def parse_hmm(path: str) -> ProfileHMM:
    with open(path) as f:
        lines = f.read().split("\n")
    it = iter(lines)
    next(it)                                           # version line
    name, M, A = "", None, 0
    alpha = np.full(127, -1, dtype=np.int32)
    for line in it:
        t = line.split()
        if not t:
            continue
        if t[0] == "NAME" and len(t) > 1:
            name = t[1]
        elif t[0] == "LENG":
            M = int(t[1])
        elif t[0] == "HMM":                            # parseAlpha, :179-201
            for i, letter in enumerate(t[1:]):
                alpha[ord(letter.upper())] = i
                alpha[ord(letter.lower())] = i
            A = len(t) - 1
            break
    if M is None or A == 0:
        raise ValueError(f"{path}: no LENG / HMM line")
    next(it)                                           # transition labels
    t = next(it).split()
    if not t or t[0] != "COMPO":
        raise ValueError(f"{path}: COMPO line required (hmmer3b_parser.h:63-75)")
    compo = np.array([math.exp(-1 * float(x)) for x in t[1:1 + A]])
    msc = np.zeros((M + 1, A))
    isc = np.zeros((M + 1, A))
    tsc = np.zeros((7, M + 1))
    max_match = np.full(M + 1, NEG_INF)
    for i in range(M + 1):
        if i > 0:
            t = next(it).split()
            for j in range(A):
                p = _p(t[1 + j])
                q = p / compo[j]
                msc[i, j] = _ln(q)
            max_match[i] = msc[i].max()
        next(it)                                       # insert emissions: forced to 0 in normalized mode (:145-147)
        t = next(it).split()
        for j in range(7):
            tsc[j, i] = _ln(_p(t[j]))
    isc[M, :] = NEG_INF                                # :170-172
    hm = ProfileHMM(name=name, M=M, A=A, alpha=alpha, compo=compo, msc=msc, isc=isc, tsc=tsc, max_match=max_match,
                    h=np.zeros((3, M + 1)))
    for i in range(M + 1):
        hm.h[0, i] = _heuristic(hm, "m", i)
        hm.h[1, i] = _heuristic(hm, "i", i)
        hm.h[2, i] = _heuristic(hm, "d", i)
    return hm


def _heuristic(hm: ProfileHMM, pre: str, state_no: int) -> float:
    """computeCostInternal, most_probable_path.h:48-118"""
    h = 0.0
    for i in range(state_no + 1, hm.M + 1):
        if pre == "m":
            mt, it_, dt = hm.tsc[MM, i - 1], hm.tsc[MI, i - 1], hm.tsc[MD, i - 1]
        elif pre == "d":
            mt, it_, dt = hm.tsc[DM, i - 1], NEG_INF, hm.tsc[DD, i - 1]
        else:
            mt, it_, dt = hm.tsc[IM, i - 1], hm.tsc[II, i - 1], NEG_INF
        best_m = max(NEG_INF, float(hm.msc[i].max()))
        mt = mt + (best_m - hm.max_match[i])
        dt = dt - hm.max_match[i]
        it_ = NEG_INF                                   # :100
        if it_ > mt and it_ > dt:
            h += it_
            pre = "i"
        elif dt > mt and dt > it_:
            h += dt
            pre = "d"
        else:
            h += mt
            pre = "m"
    return h
