"""Pins the oracle's leaf functions against the reference's own sources compiled unmodified
(oracle/_ref/libref_units.so = WordSlice.h, AlignmentCorrectnessEstimation.cpp, edlib)."""
import ctypes as C
import os
import random

import numpy as np
import pytest

from oracle import RefUnits, load_oracle_lib

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.skipif(
    not (os.path.exists(os.path.join(HERE, "..", "oracle", "_ref", "libref_units.so")) or os.path.isdir("/root/reference/src")),
    reason="reference units not built and no reference tree")


def random_column(rng, flat=False):
    """A valid column: VP & VN == 0, plus a scoreEnd."""
    style = rng.random()
    if flat or style < 0.1:
        vp, vn = 0, 0
    elif style < 0.2:
        vp, vn = (1 << 64) - 1, 0
    elif style < 0.3:
        vp, vn = 0, (1 << 64) - 1
    else:
        p = rng.choice([0.05, 0.2, 0.5])
        vp = vn = 0
        for i in range(64):
            x = rng.random()
            if x < p:
                vp |= 1 << i
            elif x < 2 * p:
                vn |= 1 << i
    return vp, vn, rng.randint(0, 300)


def perturb(rng, col):
    """A column that differs from `col` in a few rows (the common case inside the band)."""
    vp, vn, s = col
    for _ in range(rng.randint(0, 4)):
        i = rng.randrange(64)
        vp &= ~(1 << i)
        vn &= ~(1 << i)
        x = rng.random()
        if x < 0.33:
            vp |= 1 << i
        elif x < 0.66:
            vn |= 1 << i
    return vp, vn, s + rng.randint(-2, 2)


def values(col):
    vp, vn, s = col
    before = s - bin(vp).count("1") + bin(vn).count("1")
    out = [before]
    for i in range(64):
        out.append(out[-1] + ((vp >> i) & 1) - ((vn >> i) & 1))
    return out


@pytest.fixture(scope="module")
def libs():
    return load_oracle_lib(), RefUnits().lib


def test_merge_matches_reference_and_pointwise_min(libs):
    ora, ref = libs
    rng = random.Random(1)
    u64, i32 = C.c_uint64, C.c_int32
    for it in range(20000):
        a = random_column(rng)
        b = perturb(rng, a) if it % 2 else random_column(rng)
        # the reference requires |score difference| small enough for its loop; keep columns within 64 of each other
        ovp, ovn, os_ = u64(), u64(), i32()
        rvp, rvn, rs = u64(), u64(), i32()
        ora.gco_merge(*a, *b, C.byref(ovp), C.byref(ovn), C.byref(os_))
        ref.ref_merge(*a, *b, C.byref(rvp), C.byref(rvn), C.byref(rs))
        assert (ovp.value, ovn.value, os_.value) == (rvp.value, rvn.value, rs.value), (a, b)
        mins = [min(x, y) for x, y in zip(values(a), values(b))]
        assert values((ovp.value, ovn.value, os_.value)) == mins


def test_changed_min_score_get_value_before_start(libs):
    ora, ref = libs
    rng = random.Random(2)
    for it in range(20000):
        a = random_column(rng)
        b = perturb(rng, a) if it % 2 else random_column(rng)
        assert ora.gco_changed_min_score(*a, *b) == ref.ref_changed_min_score(*a, *b), (a, b)
        assert ora.gco_score_before_start(*a) == ref.ref_score_before_start(*a)
        row = rng.randrange(64)
        assert ora.gco_get_value(*a, row) == ref.ref_get_value(*a, row) == values(a)[row + 1]


def test_myers_step_is_the_cell_recurrence(libs):
    """getNextSlice against the plain DP recurrence (the reference's assertSliceCorrectness twin,
    src/GraphAlignerBitvectorCommon.h:812-826)."""
    ora, _ = libs
    rng = random.Random(3)
    u64, i32 = C.c_uint64, C.c_int32
    for _ in range(5000):
        old = random_column(rng)
        eq = rng.getrandbits(64)
        hin = rng.choice([(0, 0), (1, 0), (0, 1)])
        ovp, ovn, os_, hp, hn = u64(), u64(), i32(), u64(), u64()
        ora.gco_next_slice(eq, *old, hin[0], hin[1], C.byref(ovp), C.byref(ovn), C.byref(os_), C.byref(hp), C.byref(hn))
        o = values(old)
        n = values((ovp.value, ovn.value, os_.value))
        assert n[0] == o[0] + hin[0] - hin[1]
        for i in range(64):
            want = min(n[i] + 1, o[i + 1] + 1, o[i] + (0 if (eq >> i) & 1 else 1))
            assert n[i + 1] == want
        assert hp.value - hn.value == n[64] - o[64]


def test_correctness_hmm_matches_reference(libs):
    ora, ref = libs
    rng = random.Random(4)
    for _ in range(200):
        n = rng.randint(1, 200)
        lo, hi = rng.choice([(0, 8), (5, 20), (20, 40), (0, 70)])
        mm = np.array([rng.randint(lo, hi) for _ in range(n)], dtype=np.int32)
        oc, ow, of = np.zeros(n), np.zeros(n), np.zeros(n, dtype=np.int32)
        rc, rw, rf = np.zeros(n), np.zeros(n), np.zeros(n, dtype=np.int32)
        ora.gco_correctness_series(mm.ctypes.data, n, oc.ctypes.data, ow.ctypes.data, of.ctypes.data)
        ref.ref_correctness_series(mm.ctypes.data, n, rc.ctypes.data, rw.ctypes.data, rf.ctypes.data)
        assert np.array_equal(oc.view(np.uint64), rc.view(np.uint64))   # bit-exact doubles
        assert np.array_equal(ow.view(np.uint64), rw.view(np.uint64))
        assert np.array_equal(of, rf)


def test_edit_distance_matches_edlib(libs):
    ora, ref = libs
    rng = random.Random(5)
    for _ in range(300):
        la = rng.choice([1, 5, 63, 64, 65, 130, 500])
        a = "".join(rng.choice("ACGT") for _ in range(la))
        b = list(a)
        for _ in range(rng.randint(0, max(1, la // 5))):
            op = rng.random()
            i = rng.randrange(len(b)) if b else 0
            if op < 0.4 and b:
                b[i] = rng.choice("ACGT")
            elif op < 0.7 and b:
                del b[i]
            else:
                b.insert(i, rng.choice("ACGT"))
        b = "".join(b) or "A"
        assert ora.gco_edit_distance(a.encode(), len(a), b.encode(), len(b)) == ref.ref_edit_distance(a.encode(), len(a), b.encode(), len(b))


# ---- edlib PATH mode (src/Aligner.cpp:845) and the E-value (src/EValue.cpp) ----------------------------------------

def _mut(rng, s, rate):
    out = bytearray()
    for ch in s:
        x = rng.random()
        if x < rate / 3:
            continue
        out.append(rng.choice(b"ACGT") if x < 2 * rate / 3 else ch)
        if rng.random() < rate / 3:
            out.append(rng.choice(b"ACGT"))
    return bytes(out)


def _rand(rng, n, alphabet=b"ACGT"):
    return bytes(rng.choice(alphabet) for _ in range(n))


def _path_cases(rng):
    """(query = path letters, target = read): the shapes the chained alignment meets - similar strings, a path that covers
    only the left / right / middle part of the read (stitching keeps the longest piece), a path with a stretch the read
    lacks, unrelated strings, repeats (many equally good alignments), IUPAC letters, tiny and empty sides."""
    for n in (0, 1, 2, 63, 64, 65, 200):
        q = _rand(rng, n)
        yield q, _mut(rng, q, 0.15)
        yield q, _rand(rng, rng.randint(0, 130))
    for _ in range(30):
        q = _rand(rng, rng.randint(100, 900))
        yield q, _mut(rng, q, rng.choice([0.02, 0.1, 0.3]))
        yield q, _rand(rng, 400) + _mut(rng, q, 0.1)
        yield q, _mut(rng, q, 0.1) + _rand(rng, 350)
        yield q, _rand(rng, 150) + _mut(rng, q, 0.08) + _rand(rng, 220)
        yield q, _mut(rng, q[:len(q) // 3], 0.1) + _mut(rng, q[2 * len(q) // 3:], 0.1)
    for _ in range(10):
        unit = _rand(rng, rng.randint(1, 7))
        q = (unit * 200)[:rng.randint(50, 600)]
        yield q, _mut(rng, q, 0.1)
        yield _rand(rng, 300, b"AC"), _rand(rng, 280, b"AC")
    q = _rand(rng, 500, b"ACGTNRY")
    yield q, _mut(rng, q, 0.1)
    # above edlib's 1 MB limit: Hirschberg splits (edlib/src/edlib.cpp:1204-1212), incl. splits on the matrix border
    for qn in (1500, 2600, 4200):
        q = _rand(rng, qn)
        yield q, _mut(rng, q, 0.12)
        yield q, _rand(rng, 3000) + _mut(rng, q, 0.1)
        yield q, _mut(rng, q, 0.1) + _rand(rng, 3300)
        yield q, _rand(rng, 2000) + _mut(rng, q, 0.08) + _rand(rng, 2500)
        yield q, _mut(rng, q[:qn // 3], 0.1) + _mut(rng, q[2 * qn // 3:], 0.1)
        yield q[:300], _rand(rng, 9000)
        yield (b"ACG" * 2000)[:qn], _mut(rng, (b"ACG" * 2000)[:qn], 0.05)


def test_edit_path_matches_edlib_op_for_op():
    """oracle/edlib_path.hpp against the real edlib (oracle/_ref): same distance, same op string, same "no alignment" cases."""
    from oracle.binding import oracle_edit_path
    ref = RefUnits()
    rng = random.Random(2024)
    n = hirschberg = 0
    for q, t in _path_cases(rng):
        want_d, want_ops = ref.edit_path(q, t)
        got_d, got_ops = oracle_edit_path(q, t)
        assert got_d == want_d, (len(q), len(t))
        assert np.array_equal(got_ops, want_ops), (len(q), len(t), want_d)
        if len(want_ops):
            # an op string is an alignment: it consumes both strings and costs the distance
            assert np.sum(want_ops != 2) == len(q) and np.sum(want_ops != 1) == len(t) and np.sum(want_ops != 0) == want_d
        n += 1
        hirschberg += (20 * ((len(q) + 63) // 64) + 8) * len(t) >= 1 << 20
    assert n > 200 and hirschberg >= 15


def test_evalue_matches_reference_bitwise():
    from oracle.binding import oracle_evalue
    ref = RefUnits()
    rng = random.Random(7)
    for _ in range(300):
        ident = rng.choice([0.7, 0.5, 0.66, 0.9, rng.uniform(0.05, 0.95)])
        db, q = rng.randint(1, 10**10), rng.randint(1, 10**5)
        length = rng.randint(0, 60000)
        edits = rng.randint(0, max(1, length))
        want = ref.evalue(ident, db, q, length, edits)
        got = oracle_evalue(ident, db, q, length, edits)
        assert want.tobytes() == got.tobytes(), (ident, db, q, length, edits, want, got)
