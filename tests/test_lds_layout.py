"""The slot-minor LDS cells of the tile kernels (fft_wg.hip.h: wg_exchange, sm_thread_bit_mask) -- restated and checked on CPU.

cell(idx, slot) = g(idx) * XPB + slot with g = idx, bit 0 ^= parity(idx & M0) and (XPB = 8) bit 1 ^= parity(idx & M1), M0 / M1 the
index bits that carry thread bits 0 / 1 in the gather shapes of (L, RL).  Claims checked for every pass of L = 5 .. 10 and
XPB = 8, 16: the cells are a bijection onto [0, XPB * 2^L), the "per-thread base + constant" form the kernel computes equals
the definition, and no access conflicts under the per-instruction banking of MI355X_MICROARCH.md (ds_write_b64: groups of 16
contiguous lanes on 32 dword banks; ds_read_b64: groups of 32 lanes on 64)."""
import pytest


def bitrev(x, b):
    r = 0
    for i in range(b):
        r = (r << 1) | ((x >> i) & 1)
    return r


def rl_for(L):  # host_common.hip.h
    return 2 if L in (5, 6) else 3 if L in (7, 9) else 4


def par(x):
    return bin(x).count("1") & 1


def geometry(L, RL, P):  # WgGeom<L, RL, P>
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


def extra_cycles(cells, group, banks):
    extra = 0
    for g0 in range(0, len(cells), group):
        per = {}
        for c in set(cells[g0:g0 + group]):  # identical addresses broadcast
            for d in (2 * c, 2 * c + 1):     # an 8-byte cell covers two dword banks
                per.setdefault(d % banks, set()).add(d)
        extra += max(len(v) for v in per.values()) - 1
    return extra


@pytest.mark.parametrize("xpb", [8, 16])
@pytest.mark.parametrize("L", [5, 6, 7, 8, 9, 10, 11])
def test_slot_minor_cells_bijective_decomposable_conflict_free(L, xpb):
    RL = rl_for(L)
    R, N = 1 << RL, 1 << L
    TPT, NP = N // R, (L + RL - 1) // RL
    block = xpb * TPT
    if TPT < 4 or block > 1024:
        pytest.skip("geometry not used")
    m0, m1 = 1, 2
    for P in range(1, NP):
        inn, _ = geometry(L, RL, P)
        m0 |= inn(1, 0)
        m1 |= inn(2, 0)
    lm = 3 if xpb == 8 else 1
    low = lambda x: par(x & m0) | ((par(x & m1) << 1) if xpb == 8 else 0)  # noqa: E731
    cell = lambda idx, slot: ((idx & ~lm) | low(idx)) * xpb + slot           # noqa: E731
    assert sorted(cell(i, s) for i in range(N) for s in range(xpb)) == list(range(N * xpb))
    for P in range(NP - 1):
        _, out = geometry(L, RL, P)
        inn, _ = geometry(L, RL, P + 1)
        for fn in (out, inn):
            for tau in range(TPT):
                ti = fn(tau, 0)
                for u in range(R):
                    U = fn(0, u)
                    base = (ti & ~lm) * xpb + 3 + ((low(ti) ^ low(U)) * xpb)  # slot 3
                    assert base + (U & ~lm) * xpb == cell(fn(tau, u), 3)
        if block < 64:
            continue
        for wave in range(block // 64):
            lanes = range(wave * 64, wave * 64 + 64)
            for u in range(R):
                assert extra_cycles([cell(out(t // xpb, u), t % xpb) for t in lanes], 16, 32) == 0, ("write", L, xpb, P, u)
                assert extra_cycles([cell(inn(t // xpb, u), t % xpb) for t in lanes], 32, 64) == 0, ("read", L, xpb, P, u)


def test_two_element_grain_formula_conflicts_with_16_units():
    """The formula this replaced (built for 8 units, L = 10): with 16 units every ds_write_b64 group is 2-way conflicted --
    what SQ_LDS_BANK_CONFLICT showed on the c32 first factor (1.47e7 per launch, 0 after)."""
    L, RL, xpb = 10, 4, 16
    old = lambda idx, slot: (idx >> 2) * (4 * xpb) + ((((idx >> 1) ^ (idx >> 3)) & 1) * (2 * xpb)) + slot * 2 + ((idx ^ (idx >> 2)) & 1)  # noqa: E731
    _, out = geometry(L, RL, 0)
    lanes = range(64)
    assert extra_cycles([old(out(t // xpb, 0), t % xpb) for t in lanes], 16, 32) == 4  # one extra cycle in each of the 4 groups


# ---- the wave-split kernels (fft_split.hip.h, round 3): decomposition and XOR swizzle, restated in tools/split_model.py ---------
def _split_swizzles():
    """SplitSwizzle<LA, LB>::F as the header has them."""
    import re
    from pathlib import Path

    text = (Path(__file__).resolve().parent.parent / "kofft_amd" / "csrc" / "fft_split.hip.h").read_text()
    out = {}
    for la, lb, cols in re.findall(r"SplitSwizzle<(\d+), (\d+)> \{ static constexpr int F\[\d+\] = \{([^}]*)\}", text):
        out[(int(la), int(lb))] = [int(v) for v in cols.split(",")]
    return out


@pytest.mark.parametrize("la,lb", [(7, 6), (7, 7)])
def test_wave_split_decomposition_and_swizzle(la, lb):
    """The kernels' thread / register index maps computed on the CPU reproduce numpy's FFT, every stage uses exactly the
    reference's table entries T[k * n2] (fft.rs:839), the cell function is a bijection, XOR-linear (so that an address is
    base(thread) ^ const(register)), and none of the six LDS access shapes conflicts under the per-instruction banking."""
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))
    import split_model as sm

    assert sm.run(la, lb) < 1e-12  # also asserts the per-stage table index sets
    F = _split_swizzles()[(la, lb)]
    g = sm.Geom(la, lb)
    assert len(F) == la and all(0 <= f < (1 << lb) for f in F)
    assert sm.conflicts(g, F) == 0

    def cell(K, j):
        return (K << lb) | (j ^ sm.apply_f(F, K))

    cells = {cell(K, j) for K in range(1 << la) for j in range(1 << lb)}
    assert len(cells) == g.N and max(cells) == g.N - 1
    import random

    rnd = random.Random(5)
    for _ in range(2000):  # XOR-linearity on disjoint bit fields: what "base ^ constant" relies on
        k1, j1 = rnd.randrange(1 << la), rnd.randrange(1 << lb)
        mk, mj = rnd.randrange(1 << la), rnd.randrange(1 << lb)
        ka, kb, ja, jb = k1 & mk, k1 & ~mk, j1 & mj, j1 & ~mj
        assert cell(k1, j1) == cell(ka, ja) ^ cell(kb, jb)


@pytest.mark.parametrize("la,lb,qb0,F,G", [(7, 7, 2, [2, 9, 24, 5, 1, 4, 26], [30, 7]), (7, 7, 3, [2, 9, 24, 5, 1, 4, 26], [30, 7]),
                                        (7, 6, 2, [2, 9, 24, 30, 1, 4, 26], [30]), (6, 7, 2, [3, 9, 24, 30, 1, 4], [26, 30])])
def test_wide_wave_split_decomposition_and_swizzle(la, lb, qb0, F, G):
    """fft_split_wide.hip.h (32 points per thread, a wavefront owns 2048 points): the index maps reproduce numpy's FFT with the
    reference's per-stage table entries, cell(K, j) = K * 2^LB + (j ^ F(K) ^ G(j >> 5)) is a bijection, XOR-linear, and none
    of the six access shapes conflicts.  The F / G here are the ones SplitWideSwizzle holds."""
    import re
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root / "tools"))
    import split_model as sm

    src = (root / "kofft_amd" / "csrc" / "fft_split_wide.hip.h").read_text()
    m = re.search(r"SplitWideSwizzle<%d, %d> \{[^}]*F\[\d+\] = \{([^}]*)\};\s*static constexpr int G\[\d+\] = \{([^}]*)\};" % (la, lb), src)
    assert m and [int(v) for v in m.group(1).split(",")] == F and [int(v) for v in m.group(2).split(",")] == G
    kw = dict(rlog=5, qa0=5, qb0=qb0)
    assert sm.run(la, lb, **kw) < 1e-12
    g = sm.Geom(la, lb, **kw)
    assert sm.conflicts(g, F, G) == 0
    cells = {sm.cell_of(g, F, K, j, G) for K in range(1 << la) for j in range(1 << lb)}
    assert len(cells) == g.N and max(cells) == g.N - 1
    import random

    rnd = random.Random(6)
    for _ in range(2000):
        k1, j1 = rnd.randrange(1 << la), rnd.randrange(1 << lb)
        mk, mj = rnd.randrange(1 << la), rnd.randrange(1 << lb)
        assert sm.cell_of(g, F, k1, j1, G) == sm.cell_of(g, F, k1 & mk, j1 & mj, G) ^ sm.cell_of(g, F, k1 & ~mk, j1 & ~mj, G)


def test_wide_kernel_pairing_cells_are_base_xor_constant():
    """fft_split_wide.hip.h, rfft epilogue / irfft pre-pass (n = 32768 reals): element k = 2^LA q + K of a row lives in cell (K, q);
    thread kk handles k = 512 s + kk, i.e. K = kk mod 128, q = 4 s + q0.  Claimed: cell(k) = cell(K, q0) ^ cell(0, 4 s), and for
    K != 0 the partner m - k sits in cell(128 - K, 3 - q0) ^ cell(0, 4 (31 - s)) (q complements bit by bit); K = 0 pairs with
    q' = 128 - q instead (computed per step in the kernel)."""
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))
    import split_model as sm

    la = lb = 7
    F, G = [2, 9, 24, 5, 1, 4, 26], [30, 7]
    g = sm.Geom(la, lb, rlog=5, qa0=5, qb0=2)
    m = g.N

    def cell(K, j):
        return sm.cell_of(g, F, K, j, G)

    def cell_of_k(k):
        return cell(k & 127, k >> 7)

    for kk in range(512):
        K, q0 = kk & 127, kk >> 7
        for s in range(32):
            k = 512 * s + kk
            assert cell_of_k(k) == cell(K, q0) ^ cell(0, 4 * s)
            if K:
                assert cell_of_k(m - k) == cell(128 - K, 3 - q0) ^ cell(0, 4 * (31 - s))
            elif k:
                assert cell_of_k(m - k) == cell(0, 128 - (4 * s + q0))
