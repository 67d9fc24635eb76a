"""Tiny pure-Python parser of Snappy element streams (format_description.txt), used by the
tests to look INSIDE compressed output: which copies were taken, element sizes, literal runs."""


def parse_varint(comp):
    v, shift = 0, 0
    for i in range(5):
        b = comp[i]
        v |= (b & 0x7F) << shift
        if b < 0x80:
            return v, i + 1
        shift += 7
    raise ValueError("bad varint")


def elements(body):
    """Yields (kind, pos, length, offset, raw_bytes) for a fragment BODY (no varint);
    kind is 'lit' or 'copy'; pos is the uncompressed position the element starts at."""
    i, pos = 0, 0
    while i < len(body):
        tag = body[i]
        k = tag & 3
        if k == 0:
            ln = (tag >> 2) + 1
            hdr = 1
            if ln > 60:
                nb = ln - 60
                ln = int.from_bytes(body[i + 1:i + 1 + nb], "little") + 1
                hdr = 1 + nb
            yield ("lit", pos, ln, 0, body[i:i + hdr])
            i += hdr + ln
        elif k == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | body[i + 1]
            yield ("copy", pos, ln, off, body[i:i + 2])
            i += 2
        elif k == 2:
            ln = (tag >> 2) + 1
            off = body[i + 1] | (body[i + 2] << 8)
            yield ("copy", pos, ln, off, body[i:i + 3])
            i += 3
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(body[i + 1:i + 5], "little")
            yield ("copy", pos, ln, off, body[i:i + 5])
            i += 5
        pos += ln
