"""The mask algebra of the kernel's fixed-tile dense path (tests/tile_model.py) against the oracle, on
the CPU: walk, exit state, conflict cut, same-tile revisit and deferred emission must reproduce the
serial parse byte for byte.  (The HIP kernel itself is checked by the -m gpu tests.)"""
import pytest

import fuzzgen
import oracle_lib
import tile_model


def _check(data):
    st = {}
    got = tile_model.compress_fragment(data, stats=st)
    assert got == oracle_lib.compress_fragment(data)
    assert tile_model.compress_fragment(data, fast=False) == got
    return st


def test_corpora_blocks():
    mtg = fuzzgen.corpus("mtg")
    st = _check(mtg[:65536])
    # the path under test is really taken, with every kind of event in it
    assert st["tiles"] > 800 and st["cuts"] > 100 and st["patches"] > 50 and st["long"] > 20 and st["flagged"] > 0
    _check(mtg[7 * 65536:8 * 65536])
    _check(fuzzgen.corpus("fox"))
    _check(fuzzgen.corpus("fox")[:4096] * 16)
    _check(fuzzgen.tiled_block(fuzzgen.corpus("random_ascii"), 3))


@pytest.mark.parametrize("kind", fuzzgen.KINDS)
def test_generators(kind):
    for seed, n in ((1, 300), (2, 5000), (3, 20000), (4, 65536)):
        _check(fuzzgen.gen(kind, seed, n).tobytes())


def test_sizes_around_the_margins():
    src = fuzzgen.gen("words", 9, 70000).tobytes()
    for n in (0, 1, 14, 15, 16, 100, 127, 128, 129, 143, 144, 160, 191, 192, 193, 255, 256, 257, 300, 1000, 4095, 4096, 4097):
        _check(src[:n])


# ---- the extension prepared for a later round ("sparse scans inside the tile path"): the
# first segment of a visit follows the scan's schedule past its 32nd probe (lanes the scan skips are
# neither looked at nor inserted), later segments that would do so end the visit, and the tile path is
# entered in scans as well as at the end of copies.  Same bytes as the oracle; the kernel does not use it yet.
def _check_sparse(data):
    st = {}
    assert tile_model.compress_fragment(data, stats=st, sparse=True) == oracle_lib.compress_fragment(data)
    return st


def test_sparse_scans_in_the_tile_path_corpora():
    mtg = fuzzgen.corpus("mtg")
    st = _check_sparse(mtg[:65536])
    assert st.get("flagged", 0) == 0 and st["far_cuts"] > 0  # nothing handed back, and the cut in front of a sparse segment is taken
    st = _check_sparse(mtg[:4096] * 16)
    assert st.get("flagged", 0) == 0
    _check_sparse(fuzzgen.corpus("fox"))
    st = _check_sparse(fuzzgen.tiled_block(fuzzgen.corpus("random_ascii"), 3))
    assert st["flagged"] > 0  # steps past half a tile: the general code


@pytest.mark.parametrize("kind", fuzzgen.KINDS)
def test_sparse_scans_in_the_tile_path_generators(kind):
    for seed, n in ((1, 300), (2, 5000), (3, 20000), (4, 65536), (8, 65521)):
        _check_sparse(fuzzgen.gen(kind, seed, n).tobytes())
