#!/usr/bin/env python3
"""Conflict cycles of the PLAIN exchange layout (cell = slot * stride + pad(idx)) of fft_wg_kernel / fft_persist_kernel for
8-byte cells (c32), per (L, RL), under the per-instruction banking of MI355X_MICROARCH.md (ds_write_b64: groups of 16 contiguous
lanes on 32 dword banks; ds_read_b64: groups of 32 lanes on 64).  Lanes = [slot][tau] (threads of a transform contiguous).
Prints the extra cycles per exchange instruction for a list of candidate pad functions / slot strides.
usage: python3 tools/lds_plain_sim.py"""
import sys


def bitrev(x, b):
    r = 0
    for i in range(b):
        r = (r << 1) | ((x >> i) & 1)
    return r


def rl_for(L):
    return 2 if L in (5, 6) else 3 if L in (7, 9) else 4


def geometry(L, RL, P):
    R, N = 1 << RL, 1 << L
    TPT, NP = N // R, (L + RL - 1) // RL
    S0 = P * RL
    Q = (L - RL * (NP - 1)) if P == NP - 1 else RL
    JB = L - S0 - Q

    def in_index(tau, u):
        g, c = u >> Q, u & ((1 << Q) - 1)
        m = tau + g * TPT
        return ((m >> JB) << (L - S0)) | (c << JB) | (m & ((1 << JB) - 1))

    def out_index(tau, u):
        g, c = u >> Q, u & ((1 << Q) - 1)
        return (bitrev(c, Q) << (L - Q)) | (tau + g * TPT)

    return in_index, out_index


def extra(cells, group, banks):
    tot = 0
    for g0 in range(0, len(cells), group):
        per = {}
        for c in set(cells[g0:g0 + group]):
            for d in (2 * c, 2 * c + 1):
                per.setdefault(d % banks, set()).add(d)
        tot += max(len(v) for v in per.values()) - 1
    return tot


def evaluate(L, RL, pad, stride):
    R, N = 1 << RL, 1 << L
    TPT, NP = N // R, (L + RL - 1) // RL
    if TPT >= 64:
        lanes = [(0, t) for t in range(64)]  # one wave: 64 consecutive threads of one transform (tau0 = 0; others are shifts)
    else:
        lanes = [(s, t) for s in range(64 // TPT) for t in range(TPT)]
    w = r = n_w = n_r = 0
    for P in range(NP - 1):
        _, out = geometry(L, RL, P)
        inn, _ = geometry(L, RL, P + 1)
        for wave0 in range(0, max(TPT, 64), 64):
            for u in range(R):
                cw = [s * stride + pad(out(t + (wave0 if TPT >= 64 else 0), u)) for s, t in lanes]
                cr = [s * stride + pad(inn(t + (wave0 if TPT >= 64 else 0), u)) for s, t in lanes]
                w += extra(cw, 16, 32)
                r += extra(cr, 32, 64)
                n_w += 4   # conflict-free cost: 4 groups of 16 lanes
                n_r += 2
    return w, n_w, r, n_r


CANDS = {
    "i+(i>>4)": lambda i: i + (i >> 4),
    "i+(i>>5)": lambda i: i + (i >> 5),
    "i+(i>>3)": lambda i: i + (i >> 3),
    "i+(i>>4)+(i>>8)": lambda i: i + (i >> 4) + (i >> 8),
    "i^((i>>4)&15)": lambda i: i ^ ((i >> 4) & 15),
    "i^((i>>5)&31)": lambda i: i ^ ((i >> 5) & 31),
    "i^((i>>4)&15)^((i>>8)&15)": lambda i: i ^ ((i >> 4) & 15) ^ ((i >> 8) & 15),
    "i": lambda i: i,
}

if __name__ == "__main__":
    for L in range(5, 14):
        for RL in sorted({rl_for(L), 4 if L >= 8 else rl_for(L)}):
            N = 1 << L
            print(f"L={L} RL={RL} TPT={N >> RL}")
            for name, pad in CANDS.items():
                top = max(pad(i) for i in range(N)) + 1
                for stride in sorted({top, top + ((16 - top) % 32), N + (N >> 4) + 1}):
                    w, nw, r, nr = evaluate(L, RL, pad, stride)
                    print(f"   {name:28s} stride {stride:6d}: write +{w:4d}/{nw:4d}  read +{r:4d}/{nr:4d}")
