"""Functional model of the reference ACCELERATOR's own output ("HW-compatible mode"): sizes
(compress_size) and bytes (compress_bytes).  TEST INFRASTRUCTURE: the checker of the library's SZ_FLAG_HW_COMPAT mode
(csrc/snappy_hwcompat.hip, tests/test_gpu_hwcompat.py) -- never imported by the product; the parity target of
everything else is software Snappy (App. C).

The RTL cannot be simulated here (no JVM, empty rocket-chip); this is what its datapath does to a buffer, read
from the Scala as text, at the level of "which bytes become which elements":

  * 512-entry hash table with present bits, stored data and 16-bit offsets, cleared per job
    (HashTable.scala:11-54); the hashed word is the 4 bytes at matchB as the read aligner delivers them, first
    byte in bits 31..24 (CompressionAccelerator.scala:265 `data.bits(31, 24)` is the byte streamed out);
    hash = (word * 0x1e35a7bd >> 23) & 511 (HashTable.scala:51-53).
  * every position is hashed while at least 4 bytes remain (`remain >= 4`, CompressionAccelerator.scala:139); no
    skip heuristic, no P-1 insertion (MatchFinder.scala:42-43 TODOs); one byte per step (:315).
  * a position can only match while the finder is `looking`: on the cycle `start` fires the table is written but
    no match is reported (MatchFinder.scala:70-73) -- the first position of a job and the first position after
    every copy cannot match.
  * a match needs present && stored word == word && candidate != position (MatchFinder.scala:64-67).
  * literals are closed every 60 bytes with a one-byte tag (forceEmit, CompressionAccelerator.scala:172).
  * a copy starts at 4 bytes and grows by up to 4 per cycle to at most 64 (CopyCompress.scala:49,72,140-144);
    a longer match goes on as further elements with the same offset (`continue`, :61,80,85-89).
  * copy elements: 2 bytes for length 4..11 and offset < 2048, else 3 bytes (CopyStreamFormer, :236-250).
  * no varint preamble is written (nothing in :226-285 emits one).

What the model does NOT have is the write-back bookkeeping through which the RTL reports its length --
`rd = (storeSpAddr - 1) * 8 + finalSWPointerOffset` (:405-407) with the 16-byte stream holder, the reserved tag
slot (`emptySpot*`) and the `firstWrite` path that adds two to the stream counter when the aligner has no data
on the cycle a literal begins (:262-285) -- a function of memory timing, not of the data.  tests/test_hw_model.py
pins what that leaves: the model's size is at most the reported one, short by 2..4 bytes on the rows where
nothing else differs.
"""
KMUL = 0x1E35A7BD
TABLE = 512


def be32(d, p):
    return (d[p] << 24) | (d[p + 1] << 16) | (d[p + 2] << 8) | d[p + 3]


def hw_hash(word):
    return ((word * KMUL) >> 23) & (TABLE - 1)


def copy_element_bytes(length, offset):
    return 2 if (offset < 2048 and 4 <= length <= 11) else 3


def compress_size(data, max_copy=64, max_literal=60, follow_matches=True):
    """Size in bytes of the element stream the accelerator's datapath forms for `data` (no preamble).
    Returns (size, stats)."""
    d = bytes(data)
    n = len(d)
    present = [False] * TABLE
    t_word = [0] * TABLE
    t_off = [0] * TABLE
    out = 0
    lit = 0      # bytes of the open literal
    n_lit = n_copy = 0
    last = None  # kind of the last element formed
    p = 0
    looking = False  # False on the cycle `start` fires: the table is written, no match reported
    while p < n:
        if n - p < 4:  # the last bytes are streamed out as literal bytes
            lit += 1
            p += 1
        else:
            w = be32(d, p)
            h = hw_hash(w)
            hit = looking and present[h] and t_word[h] == w and t_off[h] != p
            cand = t_off[h]
            present[h], t_word[h], t_off[h] = True, w, p
            if not hit:
                looking = True
                lit += 1
                p += 1
            else:
                if lit:
                    out += lit + 1
                    n_lit += 1
                    lit = 0
                last = "copy"
                m = 4
                while follow_matches and p + m < n and d[cand + m] == d[p + m]:
                    m += 1
                off = p - cand
                rest = m
                while rest > 0:  # elements of at most 64 bytes, chained by `continue`
                    ln = min(rest, max_copy)
                    out += copy_element_bytes(ln, off)
                    n_copy += 1
                    rest -= ln
                p += m
                looking = False
        if lit == max_literal:
            out += lit + 1
            n_lit += 1
            lit = 0
            last = "literal"
    if lit:
        out += lit + 1
        n_lit += 1
        last = "literal"
    return out, {"literals": n_lit, "copies": n_copy, "last": last}


def compress_bytes(data, max_copy=64, max_literal=60):
    """The element stream itself (no preamble): the same walk as compress_size, emitting Snappy elements the way the
    accelerator's formers do (literal tag = (len - 1) << 2 for len <= 60; copy: CopyStreamFormer's two forms).  Put
    behind varint32(len(data)) it is a valid Snappy stream -- tests/test_hw_model.py decodes it with the oracle.
    The GPU's SZ_FLAG_HW_COMPAT mode is compared byte for byte with this."""
    d = bytes(data)
    n = len(d)
    present = [False] * TABLE
    t_word = [0] * TABLE
    t_off = [0] * TABLE
    out = bytearray()
    lit_start = 0
    p = 0
    looking = False

    def flush(upto):
        nonlocal lit_start
        ln = upto - lit_start
        if ln:
            out.append((ln - 1) << 2)
            out.extend(d[lit_start:upto])
        lit_start = upto

    while p < n:
        if n - p < 4:
            p += 1
        else:
            w = be32(d, p)
            h = hw_hash(w)
            hit = looking and present[h] and t_word[h] == w and t_off[h] != p
            cand = t_off[h]
            present[h], t_word[h], t_off[h] = True, w, p
            if not hit:
                looking = True
                p += 1
            else:
                flush(p)
                m = 4
                while p + m < n and d[cand + m] == d[p + m]:
                    m += 1
                off = p - cand
                rest = m
                while rest > 0:
                    ln = min(rest, max_copy)
                    if off < 2048 and 4 <= ln <= 11:
                        out.append(1 | ((ln - 4) << 2) | ((off >> 8) << 5))
                        out.append(off & 0xFF)
                    else:
                        out.append(2 | ((ln - 1) << 2))
                        out.append(off & 0xFF)
                        out.append(off >> 8)
                    rest -= ln
                p += m
                lit_start = p
                looking = False
        if p - lit_start == max_literal:
            flush(p)
    flush(n)
    return bytes(out)
