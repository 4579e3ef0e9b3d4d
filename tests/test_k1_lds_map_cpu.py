"""The LDS map of K1's bordered factorization (k_feat.hip, bordered_factor_lds) checked on the CPU: for every track length the
regions that are alive at the same time - operand rows of the 16-row tiles, the six sub-diagonal tiles of the factor, the exchange
tile, the corner words - never overlap and stay inside the wave's 2560 doubles.  The constants are read from the source."""
import os
import re

import pytest

SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ov_plane_amd", "csrc", "k_feat.hip")


def _constants():
    s = open(SRC).read()
    rowt = int(re.search(r"V3_ROWT\s*=\s*(\d+)", s).group(1))
    st = int(re.search(r"V3_ST\s*=\s*(\d+)", s).group(1))
    corner = int(re.search(r"V3_CORNER\s*=\s*(\d+)", s).group(1))
    m = re.search(r"return ct == 0 \? \(rt == 1 \? (\d+) : \(rt == 2 \? (\d+) : (\d+)\)\) : \(ct == 1 \? \(rt == 2 \? (\d+) : (\d+)\) : (\d+)\);", s)
    t10, t20, t30, t21, t31, t32 = (int(x) for x in m.groups())
    ltile = {(1, 0): t10, (2, 0): t20, (3, 0): t30, (2, 1): t21, (3, 1): t31, (3, 2): t32}
    return rowt, st, corner, ltile


def _row_region(rowt, tile, row_in_tile):
    """[J | C | E] of one measurement row: three pieces (offsets as in feat_body / bordered_factor_lds)."""
    base = rowt * tile
    off_c, off_e = (72, 240) if tile == 3 else (96, 320)
    return [(base + 6 * row_in_tile, 6), (base + off_c + 14 * row_in_tile, 14), (base + off_e + 14 * row_in_tile, 14)]


def _overlap(a, b):
    return a[0] < b[0] + b[1] and b[0] < a[0] + a[1]


@pytest.mark.parametrize("m", list(range(2, 31)))
def test_regions_alive_together_never_overlap(m):
    rowt, st, corner, ltile = _constants()
    n = 2 * m
    nb4 = n + 4
    nblk = (nb4 + 15) // 16
    exchange = (st, 256)
    corner_reg = (corner, 16)
    # phase A2: all operand rows + P_cc (196 doubles in the exchange area)
    rows = {r: _row_region(rowt, r >> 4, r & 15) for r in range(n)}
    flat = [(reg, "row %d" % r) for r, regs in rows.items() for reg in regs]
    for i in range(len(flat)):
        assert flat[i][0][0] >= 0 and flat[i][0][0] + flat[i][0][1] <= st, flat[i]
        for j in range(i + 1, len(flat)):
            assert not _overlap(flat[i][0], flat[j][0]), (flat[i], flat[j])
    assert st + 196 <= 2560
    published = []  # L tiles written so far
    corner_written = False
    for jb in range(nblk):
        j0 = 16 * jb
        # alive while block jb is built and factorized: operand rows of tiles >= jb, the tiles published by earlier blocks, the exchange
        # tile, the corner words once a corner column has been finished
        alive = [(reg, "row %d" % r) for r, regs in rows.items() if (r >> 4) >= jb for reg in regs]
        alive += [((ltile[t], 256), "L%s" % (t,)) for t in published]
        alive.append((exchange, "exchange"))
        if corner_written:
            alive.append((corner_reg, "corner"))
        for i in range(len(alive)):
            lo, ln = alive[i][0]
            assert lo >= 0 and lo + ln <= 2560, alive[i]
            for j in range(i + 1, len(alive)):
                assert not _overlap(alive[i][0], alive[j][0]), (m, jb, alive[i], alive[j])
        # end of the block: its operand rows are dead; the sub-diagonal tiles of its column and (if it holds corner columns) the corner
        # words are written - they must not touch anything that later blocks still read
        new_tiles = [(rt, jb) for rt in range(jb + 1, nblk)] if jb + 1 < nblk else []
        later = [(reg, "row %d" % r) for r, regs in rows.items() if (r >> 4) > jb for reg in regs]
        later += [((ltile[t], 256), "L%s" % (t,)) for t in published]
        later.append((exchange, "exchange"))
        writes = [((ltile[t], 256), "L%s" % (t,)) for t in new_tiles]
        if j0 + 16 > n:
            writes.append((corner_reg, "corner"))
            corner_written = True
        for w in writes:
            assert w[0][0] >= 0 and w[0][0] + w[0][1] <= 2560, w
            for a in later:
                assert not _overlap(w[0], a[0]), (m, jb, w, a)
        for i in range(len(writes)):
            for j in range(i + 1, len(writes)):
                assert not _overlap(writes[i][0], writes[j][0]), (writes[i], writes[j])
        published += new_tiles


def test_budget_is_the_one_the_source_states():
    rowt, st, corner, ltile = _constants()
    # 60 rows of 34 doubles end where the exchange tile starts; the last L tile ends inside the wave's share of the workgroup's LDS
    assert 3 * rowt + 12 * 34 == st
    assert max(ltile.values()) + 256 <= 2560
    assert 160 * 1024 // 8 // 8 == 2560
