"""k_chol2's dx = L0 y (k_chol2.hip, round 5): wave w owns rows w, w + 12, ...; the rows are taken in static batches (first row index
JLO, CNT rows, K pieces of 64 16-byte pairs each).  Checked here on the CPU from the batch list in the source: every row below the
kernel's limit is owned by exactly one (wave, batch, slot), and a batch's K pieces cover the row's non-zeros."""
import os
import re

SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ov_plane_amd", "csrc", "k_chol2.hip")


def _batches():
    s = open(SRC).read()
    pat = r"batch\(std::integral_constant<int, (\d+)>\{\}, std::integral_constant<int, (\d+)>\{\}, std::integral_constant<int, (\d+)>\{\}\);"
    b = [tuple(int(x) for x in m) for m in re.findall(pat, s)]
    waves = int(re.search(r"C2_WAVES\s*=\s*(\d+)", s).group(1))
    return b, waves


def test_every_row_is_owned_once_and_covered():
    batches, waves = _batches()
    assert waves == 12 and len(batches) >= 3
    limit = 288  # ovp_chol2_max_n() + 1 rows at most (n_full <= 288)
    owner = {}
    for (jlo, cnt, k) in batches:
        assert cnt <= 16  # the transposed reduction of a batch holds at most 16 partial sums
        for w in range(waves):
            for j in range(cnt):
                row = w + waves * (jlo + j)
                assert row not in owner, (row, owner.get(row), (jlo, cnt, k))
                owner[row] = (jlo, cnt, k)
                # a row's non-zeros that meet y: columns 0..min(row, n - 1): at most row + 1 doubles = ceil((row + 2) / 2) pairs
                if row < limit:
                    assert 64 * k >= (row + 2) // 2, (row, k)
    assert all(r in owner for r in range(limit))
    # batches are contiguous in the slot index (no gap a wave would skip)
    slots = sorted((jlo, cnt) for jlo, cnt, _ in batches)
    pos = 0
    for jlo, cnt in slots:
        assert jlo == pos
        pos += cnt
    assert waves * pos >= limit
