#!/usr/bin/env python3
"""CPU model of fft_split_persist_kernel (kofft_amd/csrc/fft_split.hip.h): one workgroup per transform of N = 2^(LA+LB)
points, 16 points per thread, every WAVEFRONT owning 1024 points.

  phase A  stages 0 .. LA-1   : 2^LB column transforms of 2^LA points (element i = a * 2^LB + j); wavefront w owns the
                                CA = 1024 / 2^LA adjacent columns j = w*CA + x.  Register passes of 4 and LA-4 stages with a
                                WAVE-LOCAL exchange between them (no s_barrier: only this wavefront touches those cells).
  block exchange (the one s_barrier per transform): cell (K, j), K = phase A's output frequency
  phase B  stages LA .. L-1   : 2^LA row transforms of 2^LB points with frequency prefix K; wavefront w owns the
                                RB = 1024 / 2^LB adjacent rows K = w*RB + y.  Passes of 4 and LB-4 stages, wave-local exchange.
  output o = q * 2^LA + K.

This file checks, for every (LA, LB) the kernel is instantiated with:
  1. the index algebra, by running the decomposition numerically against numpy's FFT;
  2. that the table indices equal the ones the reference's stage loop uses (fft.rs:836-898: stage s, group k -> T[k * n2]);
  3. an XOR swizzle F (cell(K, j) = K * 2^LB + (j ^ F(K)), F linear over GF(2)) that makes all six LDS access shapes
     conflict-free under the per-instruction banking (ds_write_b64: 16 lanes on 32 dword banks; ds_read_b64: 32 on 64).
usage: python3 tools/split_model.py [LA LB]"""
import itertools
import sys

import numpy as np


def bitrev(x, b):
    r = 0
    for i in range(b):
        r = (r << 1) | ((x >> i) & 1)
    return r


class Geom:
    def __init__(self, LA, LB):
        self.LA, self.LB, self.L = LA, LB, LA + LB
        self.N = 1 << self.L
        self.W = self.N // 1024
        self.QA1, self.QB1 = LA - 4, LB - 4
        assert 1 <= self.QA1 <= 4 and 1 <= self.QB1 <= 4
        self.CA, self.RB = 1024 >> LA, 1024 >> LB   # columns / rows per wavefront
        self.TA, self.TB = 1 << (LA - 4), 1 << (LB - 4)  # threads per column / per row

    # ---- phase A, lane = ja * CA + x
    def a_lane(self, lane):
        return lane // self.CA, lane % self.CA  # (ja, x)

    def a0_in(self, w, lane, c):      # global element index loaded into register c
        ja, x = self.a_lane(lane)
        return ((c << (self.LA - 4)) | ja) << self.LB | (w * self.CA + x)

    def a0_out(self, lane, c):        # local index (within the column transform) after pass A0
        ja, _ = self.a_lane(lane)
        return (bitrev(c, 4) << (self.LA - 4)) | ja

    def a1_in(self, lane, u):         # local index gathered into register u = (g, c') of pass A1
        kk, _ = self.a_lane(lane)
        Q = self.QA1
        g, c = u >> Q, u & ((1 << Q) - 1)
        k = kk + g * self.TA
        return (k << Q) | c, k

    def a1_out(self, lane, u):        # K: the column transform's output frequency
        kk, _ = self.a_lane(lane)
        Q = self.QA1
        g, c = u >> Q, u & ((1 << Q) - 1)
        return (bitrev(c, Q) << 4) | (kk + g * self.TA)

    # ---- phase B, lane = jb * RB + y
    def b_lane(self, lane):
        return lane // self.RB, lane % self.RB  # (jb, y)

    def b0_in(self, lane, c):         # j within the row
        jb, _ = self.b_lane(lane)
        return (c << (self.LB - 4)) | jb

    def b0_out(self, lane, c):
        jb, _ = self.b_lane(lane)
        return (bitrev(c, 4) << (self.LB - 4)) | jb

    def b1_in(self, lane, u):
        kb, _ = self.b_lane(lane)
        Q = self.QB1
        g, c = u >> Q, u & ((1 << Q) - 1)
        k = kb + g * self.TB
        return (k << Q) | c, k

    def b1_out(self, w, lane, u):     # global output index
        kb, y = self.b_lane(lane)
        Q = self.QB1
        g, c = u >> Q, u & ((1 << Q) - 1)
        q = (bitrev(c, Q) << 4) | (kb + g * self.TB)
        return (q << self.LA) | (w * self.RB + y)


def reg_pass(v, Lsub, S0, Q, k, tw_index, used):
    """reg_pass of fft_device.hip.h on 2^Q values; tw_index(idx_local, s_local) -> table index; returns nothing (in place)."""
    for t in range(Q):
        pos = Q - 1 - t
        for h in range(1 << t):
            idx = (k << (Lsub - 1 - S0 - t)) + (bitrev(h, t) << (Lsub - 1 - t))
            ti = tw_index(idx, S0 + t)
            used.append((S0 + t, ti))
            for lo in range(1 << pos):
                c = (h << (pos + 1)) | lo
                e, o = v[c], v[c | (1 << pos)]
                tt = o * TW[ti]
                v[c], v[c | (1 << pos)] = e + tt, e - tt


TW = None


def run(LA, LB, seed=0):
    global TW
    g = Geom(LA, LB)
    N, L = g.N, g.L
    TW = np.exp(-2j * np.pi * np.arange(N // 2) / N)
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    lds = {}
    out = np.zeros(N, complex)
    used_a, used_b = [], []
    # phase A
    for w in range(g.W):
        regs = {}
        for lane in range(64):
            v = [x[g.a0_in(w, lane, c)] for c in range(16)]
            reg_pass(v, LA, 0, 4, 0, lambda idx, s: idx << LB, used_a)
            for c in range(16):
                lds[("A", w, g.a0_out(lane, c), g.a_lane(lane)[1])] = v[c]
        for lane in range(64):
            _, xcol = g.a_lane(lane)
            v = [None] * 16
            ks = [None] * 16
            for u in range(16):
                loc, k = g.a1_in(lane, u)
                v[u] = lds[("A", w, loc, xcol)]
                ks[u] = k
            Q = g.QA1
            for grp in range(16 >> Q):
                sub = v[grp << Q:(grp + 1) << Q]
                reg_pass(sub, LA, 4, Q, ks[grp << Q], lambda idx, s: idx << LB, used_a)
                v[grp << Q:(grp + 1) << Q] = sub
            for u in range(16):
                lds[("X", g.a1_out(lane, u), w * g.CA + xcol)] = v[u]
    # phase B
    for w in range(g.W):
        for lane in range(64):
            _, y = g.b_lane(lane)
            K = w * g.RB + y
            v = [lds[("X", K, g.b0_in(lane, c))] for c in range(16)]
            reg_pass(v, LB, 0, 4, 0, lambda idx, s: (idx << LA) + (K << (LB - 1 - s)), used_b)
            for c in range(16):
                lds[("B", K, g.b0_out(lane, c))] = v[c]
        for lane in range(64):
            _, y = g.b_lane(lane)
            K = w * g.RB + y
            v, ks = [None] * 16, [None] * 16
            for u in range(16):
                loc, k = g.b1_in(lane, u)
                v[u] = lds[("B", K, loc)]
                ks[u] = k
            Q = g.QB1
            for grp in range(16 >> Q):
                sub = v[grp << Q:(grp + 1) << Q]
                reg_pass(sub, LB, 4, Q, ks[grp << Q], lambda idx, s: (idx << LA) + (K << (LB - 1 - s)), used_b)
                v[grp << Q:(grp + 1) << Q] = sub
            for u in range(16):
                out[g.b1_out(w, lane, u)] = v[u]
    want = np.fft.fft(x)
    err = np.linalg.norm(out - want) / np.linalg.norm(want)
    # table indices: stage s of the reference's loop uses T[k * n2], k < 2^s, n2 = N / 2^(s+1)
    stages = {}
    for s, ti in used_a:
        stages.setdefault(s, set()).add(ti)
    for s, ti in used_b:
        stages.setdefault(s + LA, set()).add(ti)
    for s in range(L):
        n2 = N >> (s + 1)
        assert stages[s] == {k * n2 for k in range(1 << s)}, (s, sorted(stages[s])[:8])
    return err


# ---- LDS shapes and the swizzle search ---------------------------------------------------------------------------------
def shapes(g):
    """(name, kind, [per-lane list of (K, j)] for register 0).  Conflicts depend on which index bits the lanes of a group
    differ in, and the register only adds a constant (XOR-linear cells): one register per shape is enough -- all are checked anyway."""
    out = []
    for reg in range(16):
        out.append((f"A0 scatter r{reg}", "w", [(g.a0_out(l, reg), g.a_lane(l)[1]) for l in range(64)]))
        out.append((f"A1 gather r{reg}", "r", [(g.a1_in(l, reg)[0], g.a_lane(l)[1]) for l in range(64)]))
        out.append((f"A1 scatter r{reg}", "w", [(g.a1_out(l, reg), g.a_lane(l)[1]) for l in range(64)]))
        out.append((f"B0 gather r{reg}", "r", [(g.b_lane(l)[1], g.b0_in(l, reg)) for l in range(64)]))
        out.append((f"B0 scatter r{reg}", "w", [(g.b_lane(l)[1], g.b0_out(l, reg)) for l in range(64)]))
        out.append((f"B1 gather r{reg}", "r", [(g.b_lane(l)[1], g.b1_in(l, reg)[0]) for l in range(64)]))
    return out


def apply_f(F, K):
    r = 0
    i = 0
    while K:
        if K & 1:
            r ^= F[i]
        K >>= 1
        i += 1
    return r


def conflicts(g, F):
    total = 0
    for name, kind, cells in shapes(g):
        group, banks = (16, 16) if kind == "w" else (32, 32)
        for g0 in range(0, 64, group):
            seen = {}
            for K, j in cells[g0:g0 + group]:
                cell = (K << g.LB) | (j ^ apply_f(F, K))
                seen.setdefault(cell % banks, set()).add(cell)
            total += max(len(v) for v in seen.values()) - 1
    return total


def search(g, tries=20000, seed=1):
    rng = np.random.default_rng(seed)
    nb = g.LA
    best = None
    # only the low 5 bits of j decide a bank; keep F inside the j field
    mask = (1 << min(g.LB, 5)) - 1
    cand0 = [0] * nb
    c0 = conflicts(g, cand0)
    if c0 == 0:
        return cand0, 0
    best = (c0, cand0)
    for _ in range(tries):
        F = [int(rng.integers(0, mask + 1)) for _ in range(nb)]
        c = conflicts(g, F)
        if c < best[0]:
            best = (c, F)
            # local improvement: greedy single-column changes
            improved = True
            while improved and best[0] > 0:
                improved = False
                for i, val in itertools.product(range(nb), range(mask + 1)):
                    F2 = list(best[1])
                    F2[i] = val
                    c2 = conflicts(g, F2)
                    if c2 < best[0]:
                        best = (c2, F2)
                        improved = True
            if best[0] == 0:
                break
    return best[1], best[0]


if __name__ == "__main__":
    pairs = [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else [(6, 6), (7, 6), (7, 7), (8, 6)]
    for LA, LB in pairs:
        err = run(LA, LB)
        g = Geom(LA, LB)
        F, c = search(g)
        print(f"LA={LA} LB={LB} N={g.N} waves={g.W}: rel err vs numpy {err:.2e}; table indices = reference's; swizzle F={F} "
              f"(sparse: {[hex(f) for f in F]}) residual conflict cycles {c}")
