"""CPU restatement of the chunk-cell-major pair-row ids (calls of >= 3 views, View::pairchunks): the sort kernel's
formula (binning.hip::cell_lists_from_masks - prefix over the 64-record chunks of a tile list, prefix over the cells of
a chunk, rank inside the cell) and the pair reduction's (render_bwd.hip::hgs_k_pair_reduce_ch - ballots over the masks of
a chunk's records, cell after cell) must name the same row for every (entry, cell) pair, the rows must tile [0, pairs)
and a cell's rows inside a chunk must be consecutive in list order (what makes both kernels stream)."""
import numpy as np


def rec_tag(mask, k, n, chunk_rows):
    """hgs_common.h::hgs_rec_tag"""
    left = n - (k & ~63)
    return (mask & 0xffff) | ((k & 63) << 16) | ((min(left, 64) - 1) << 22) | ((1 << 28) if chunk_rows else 0)


def sort_ids(masks, pair_base=0):
    """row id of (list position k, cell c) as the sort kernel computes it"""
    n = len(masks)
    ids = {}
    chunk_pairs = [sum(bin(int(m)).count("1") for m in masks[c0:c0 + 64]) for c0 in range(0, n, 64)]
    chunk_first = np.concatenate([[0], np.cumsum(chunk_pairs)])            # S.tab[ch][16]: exclusive prefix over the chunks
    for ch, c0 in enumerate(range(0, n, 64)):
        chunk = masks[c0:c0 + 64]
        cp = 0
        for c in range(16):
            ex = 0                                                        # lanes before this one with bit c
            for lane, m in enumerate(chunk):
                if (int(m) >> c) & 1:
                    ids[(c0 + lane, c)] = pair_base + int(chunk_first[ch]) + cp + ex
                    ex += 1
            cp += ex                                                      # tot[c]: records of the chunk that reach cell c
    entpair_y = [pair_base + int(chunk_first[k // 64]) for k in range(n)]     # what the sort leaves per record: its chunk's first row
    return ids, int(chunk_first[-1]), entpair_y


def reduce_ids(masks, entpair_y):
    """the same ids as the reduction finds them from what the sort left per record (tag word, entpair.y): a wave per
    window of 64 records takes the chunks that START in it"""
    n = len(masks)
    tags = [rec_tag(int(m), k, n, True) for k, m in enumerate(masks)]
    ids = {}
    handled = set()
    for w0 in range(0, n, 64):
        starts = [p for p in range(w0, min(n, w0 + 64)) if ((tags[p] >> 16) & 63) == 0]
        for p0 in starts:
            C = ((tags[p0] >> 22) & 63) + 1
            assert (tags[p0] >> 28) & 1
            cp = 0
            for c in range(16):
                rank = 0
                for l in range(C):
                    assert p0 + l not in handled or c > 0
                    if (tags[p0 + l] >> c) & 1:
                        ids[(p0 + l, c)] = entpair_y[p0 + l] + cp + rank
                        rank += 1
                cp += rank
            handled.update(range(p0, p0 + C))
    assert handled == set(range(n))                                       # every record belongs to exactly one chunk
    return ids


def test_chunk_cell_major_ids_tile_the_rows_and_agree_between_sort_and_reduce():
    rng = np.random.default_rng(5)
    for n in (1, 5, 63, 64, 65, 127, 128, 200, 437, 1000):
        for density in (0.05, 0.27, 0.9):
            masks = np.zeros(n, dtype=np.uint32)
            for c in range(16):
                masks |= (rng.random(n) < density).astype(np.uint32) << c
            if n > 3:
                masks[1] = 0                                              # an entry without pairs
                masks[2] = 0xffff                                         # one that reaches every cell
            a, pairs, ey = sort_ids(masks, pair_base=1000)
            assert sorted(a.values()) == list(range(1000, 1000 + pairs))                       # a bijection onto the tile's rows
            b = reduce_ids(masks, ey)
            assert a == b
            # the backward's side: consecutive list entries of a cell inside one chunk own consecutive rows
            for c in range(16):
                ks = [k for k in range(n) if (int(masks[k]) >> c) & 1]
                for k0, k1 in zip(ks, ks[1:]):
                    if k0 // 64 == k1 // 64:
                        assert a[(k1, c)] == a[(k0, c)] + 1
            # the reduction's side: the rows of an entry ascend with the cell (summation order = cell order)
            for k in range(n):
                rows = [a[(k, c)] for c in range(16) if (int(masks[k]) >> c) & 1]
                assert rows == sorted(rows)


def test_record_tag_fields():
    for n in (1, 64, 65, 130, 4096):
        for k in (0, n // 2, n - 1):
            t = rec_tag(0xbeef, k, n, True)
            assert t & 0xffff == 0xbeef and (t >> 16) & 63 == k % 64 and (t >> 28) & 1 == 1
            assert ((t >> 22) & 63) + 1 == min(64, n - (k // 64) * 64)
            assert (rec_tag(0, k, n, False) >> 28) & 1 == 0
