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
    """rlog = log2(points per thread): 4 (the kernels of fft_split.hip.h up to round 3) or 5 (the wide kernel: a wavefront owns
    2048 points); qa0 / qb0 = stages of the first register pass of each phase (the second pass takes the rest)."""

    def __init__(self, LA, LB, rlog=4, qa0=None, qb0=None):
        self.LA, self.LB, self.L = LA, LB, LA + LB
        self.N = 1 << self.L
        self.RLOG, self.R = rlog, 1 << rlog
        self.PW = 64 * self.R                 # points per wavefront
        self.W = self.N // self.PW
        self.QA0 = rlog if qa0 is None else qa0
        self.QB0 = rlog if qb0 is None else qb0
        self.QA1, self.QB1 = LA - self.QA0, LB - self.QB0
        assert 1 <= self.QA1 <= rlog and 1 <= self.QB1 <= rlog and self.QA0 <= rlog and self.QB0 <= rlog
        self.CA, self.RB = self.PW >> LA, self.PW >> LB   # columns / rows per wavefront
        self.TA, self.TB = (1 << LA) // self.R, (1 << LB) // self.R  # threads per column / per row

    # ---- phase A, lane = ja * CA + x.  Pass A0: groups g of 2^QA0 registers (one group when QA0 == RLOG)
    def a_lane(self, lane):
        return lane // self.CA, lane % self.CA  # (ja, x)

    def _split(self, u, Q):
        return u >> Q, u & ((1 << Q) - 1)

    def a0_pos(self, lane, u):        # position m of register u's group inside the column: element = c * 2^(LA-QA0) + m
        ja, _ = self.a_lane(lane)
        g, _ = self._split(u, self.QA0)
        return ja + g * self.TA

    def a0_in(self, w, lane, u):      # global element index loaded into register u
        _, x = self.a_lane(lane)
        _, c = self._split(u, self.QA0)
        return ((c << (self.LA - self.QA0)) | self.a0_pos(lane, u)) << self.LB | (w * self.CA + x)

    def a0_out(self, lane, u):        # local index (within the column transform) after pass A0
        _, c = self._split(u, self.QA0)
        return (bitrev(c, self.QA0) << (self.LA - self.QA0)) | self.a0_pos(lane, u)

    def a1_in(self, lane, u):         # local index gathered into register u = (g, c') of pass A1
        kk, _ = self.a_lane(lane)
        Q = self.QA1
        g, c = self._split(u, Q)
        k = kk + g * self.TA
        return (k << Q) | c, k

    def a1_out(self, lane, u):        # K: the column transform's output frequency
        kk, _ = self.a_lane(lane)
        Q = self.QA1
        g, c = self._split(u, Q)
        return (bitrev(c, Q) << self.QA0) | (kk + g * self.TA)

    # ---- phase B, lane = jb * RB + y
    def b_lane(self, lane):
        return lane // self.RB, lane % self.RB  # (jb, y)

    def b0_pos(self, lane, u):
        jb, _ = self.b_lane(lane)
        g, _ = self._split(u, self.QB0)
        return jb + g * self.TB

    def b0_in(self, lane, u):         # j within the row
        _, c = self._split(u, self.QB0)
        return (c << (self.LB - self.QB0)) | self.b0_pos(lane, u)

    def b0_out(self, lane, u):
        _, c = self._split(u, self.QB0)
        return (bitrev(c, self.QB0) << (self.LB - self.QB0)) | self.b0_pos(lane, u)

    def b1_in(self, lane, u):
        kb, _ = self.b_lane(lane)
        Q = self.QB1
        g, c = self._split(u, Q)
        k = kb + g * self.TB
        return (k << Q) | c, k

    def b1_out(self, w, lane, u):     # global output index
        kb, y = self.b_lane(lane)
        Q = self.QB1
        g, c = self._split(u, Q)
        q = (bitrev(c, Q) << self.QB0) | (kb + g * self.TB)
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


def run(LA, LB, seed=0, **kw):
    global TW
    g = Geom(LA, LB, **kw)
    N, L, R = g.N, g.L, g.R
    TW = np.exp(-2j * np.pi * np.arange(N // 2) / N)
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    lds = {}
    out = np.zeros(N, complex)
    used_a, used_b = [], []

    def passes(v, Lsub, S0, Q, ks, tw_index, used):
        for grp in range(R >> Q):
            sub = v[grp << Q:(grp + 1) << Q]
            reg_pass(sub, Lsub, S0, Q, ks[grp << Q], tw_index, used)
            v[grp << Q:(grp + 1) << Q] = sub

    # phase A
    for w in range(g.W):
        for lane in range(64):
            v = [x[g.a0_in(w, lane, u)] for u in range(R)]
            # pass A0 acts on stages 0 .. QA0-1: the group's "k" is 0 (all twiddle indices come from the stage-local h)
            passes(v, LA, 0, g.QA0, [0] * R, lambda idx, s: idx << LB, used_a)
            for u in range(R):
                lds[("A", w, g.a0_out(lane, u), g.a_lane(lane)[1])] = v[u]
        for lane in range(64):
            _, xcol = g.a_lane(lane)
            v, ks = [None] * R, [None] * R
            for u in range(R):
                loc, k = g.a1_in(lane, u)
                v[u] = lds[("A", w, loc, xcol)]
                ks[u] = k
            passes(v, LA, g.QA0, g.QA1, ks, lambda idx, s: idx << LB, used_a)
            for u in range(R):
                lds[("X", g.a1_out(lane, u), w * g.CA + xcol)] = v[u]
    # phase B
    for w in range(g.W):
        for lane in range(64):
            _, y = g.b_lane(lane)
            K = w * g.RB + y
            v = [lds[("X", K, g.b0_in(lane, u))] for u in range(R)]
            passes(v, LB, 0, g.QB0, [0] * R, lambda idx, s: (idx << LA) + (K << (LB - 1 - s)), used_b)
            for u in range(R):
                lds[("B", K, g.b0_out(lane, u))] = v[u]
        for lane in range(64):
            _, y = g.b_lane(lane)
            K = w * g.RB + y
            v, ks = [None] * R, [None] * R
            for u in range(R):
                loc, k = g.b1_in(lane, u)
                v[u] = lds[("B", K, loc)]
                ks[u] = k
            passes(v, LB, g.QB0, g.QB1, ks, lambda idx, s: (idx << LA) + (K << (LB - 1 - s)), used_b)
            for u in range(R):
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
    for reg in range(g.R):
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


def cell_of(g, F, K, j, G=()):
    """cell(K, j) = K * 2^LB + (j ^ F(K) ^ G(j >> 5)): F, G linear over GF(2) with values in the low five bits of j (the bank bits).
    G (wide kernel only) folds the HIGH bits of j in: two lanes of a read group that differ only in bit 5 of j share a bank otherwise."""
    return (K << g.LB) | (j ^ apply_f(F, K) ^ (apply_f(G, j >> 5) if G else 0))


CELL_BYTES = 8  # 8: one complex f32 / one f64 part per cell (ds_*_b64); 4: one f32 part per cell (ds_*_b32, fft_regfile.hip.h)


def conflicts(g, F, G=()):
    """Extra LDS cycles over all access shapes.  8-byte cells: ds_write_b64 serves 4 groups of 16 contiguous lanes on 16 cells' worth of
    banks, ds_read_b64 2 groups of 32 on 32.  4-byte cells: ds_write_b32 / ds_read_b32 serve 2 groups of 32 lanes on 32 banks, and a
    2-way conflict on the write costs nothing (MI355X_MICROARCH.md, LDS: the store's register transfer takes 4 cycles anyway)."""
    total = 0
    for name, kind, cells in shapes(g):
        if CELL_BYTES == 8:
            group, banks, free = (16, 16, 1) if kind == "w" else (32, 32, 1)
        else:
            group, banks, free = (32, 32, 2) if kind == "w" else (32, 32, 1)
        for g0 in range(0, 64, group):
            seen = {}
            for K, j in cells[g0:g0 + group]:
                cell = cell_of(g, F, K, j, G)
                seen.setdefault(cell % banks, set()).add(cell)
            total += max(0, max(len(v) for v in seen.values()) - free)
    return total


def search(g, tries=20000, seed=1, with_g=False):
    rng = np.random.default_rng(seed)
    nb = g.LA
    ng = max(0, g.LB - 5) if with_g else 0
    # only the low 5 bits of j decide a bank; keep F (and G) inside them
    mask = (1 << min(g.LB, 5)) - 1
    cand0 = [0] * (nb + ng)
    c0 = conflicts(g, cand0[:nb], cand0[nb:])
    if c0 == 0:
        return (cand0[:nb], cand0[nb:], 0) if with_g else (cand0, 0)
    best = (c0, cand0)
    for _ in range(tries):
        F = [int(rng.integers(0, mask + 1)) for _ in range(nb + ng)]
        c = conflicts(g, F[:nb], F[nb:])
        if c < best[0]:
            best = (c, F)
            # local improvement: greedy single-column changes
            improved = True
            while improved and best[0] > 0:
                improved = False
                for i, val in itertools.product(range(nb + ng), range(mask + 1)):
                    F2 = list(best[1])
                    F2[i] = val
                    c2 = conflicts(g, F2[:nb], F2[nb:])
                    if c2 < best[0]:
                        best = (c2, F2)
                        improved = True
            if best[0] == 0:
                break
    if with_g:
        return best[1][:nb], best[1][nb:], best[0]
    return best[1], best[0]


if __name__ == "__main__":
    pairs = [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else [(6, 6), (7, 6), (7, 7), (8, 6)]
    for LA, LB in pairs:
        err = run(LA, LB)
        g = Geom(LA, LB)
        F, c = search(g)
        print(f"LA={LA} LB={LB} N={g.N} waves={g.W}: rel err vs numpy {err:.2e}; table indices = reference's; swizzle F={F} "
              f"(sparse: {[hex(f) for f in F]}) residual conflict cycles {c}")
