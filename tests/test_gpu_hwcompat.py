"""SURVEY.md 8f-4 on the GPU: SZ_FLAG_HW_COMPAT -- the element stream of the reference accelerator's own datapath
(csrc/snappy_hwcompat.hip) against its CPU model (tests/hw_model.py: the same walk, read from the RTL), byte for byte,
through the C ABI; the slots decode with the software decoder; and the sizes meet the column the reference holds
(benchmark/hw_results.csv) the way tests/test_hw_model.py pins it."""
import json
import os

import numpy as np
import pytest

import fuzzgen
import hw_model
import oracle_lib
from gpu_util import compress_many, pkg

pytestmark = pytest.mark.gpu
G = fuzzgen.GOLDEN_DIR


def _varint(n):
    out = bytearray()
    while n >= 0x80:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    out.append(n)
    return bytes(out)


def _rows():
    with open(os.path.join(G, "hw_results_sizes.json")) as f:
        spec = json.load(f)
    for r in spec["rows"]:
        name = "%s-%d.txt" % (spec["file_prefix"][r["type"]], r["length"])
        with open(os.path.join(G, "data", "benchmark-data", name), "rb") as f:
            yield r["type"], r["length"], f.read(), r["compressed_length"]


def test_the_reference_benchmark_files_in_the_accelerators_own_format():
    rows = list(_rows())
    got = compress_many([d for _, _, d, _ in rows], varint=False, hw_compat=True)
    exact = 0
    for (kind, n, data, want), g in zip(rows, got):
        assert g == hw_model.compress_bytes(data), (kind, n)
        assert oracle_lib.uncompress(_varint(n) + g) == data
        last_is_literal = hw_model.compress_size(data)[1]["last"] == "literal"
        exact += len(g) + len(_varint(n)) + (1 if last_is_literal else 0) == want
    assert exact == 31  # (tests/test_hw_model.py names the five rows the datapath model does not explain)
    # with the preamble every slot is a raw stream of its own
    got_v = compress_many([d for _, _, d, _ in rows], varint=True, hw_compat=True)
    for (kind, n, data, want), g, gv in zip(rows, got, got_v):
        assert gv == _varint(n) + g


def test_generated_blocks_of_every_kind_and_size():
    rs = np.random.RandomState(77)
    edges = [0, 1, 3, 4, 5, 59, 60, 61, 63, 64, 65, 67, 68, 119, 120, 121, 127, 128, 129, 4095, 4096, 65535, 65536]
    blocks = []
    for i in range(400):
        k = fuzzgen.KINDS[i % len(fuzzgen.KINDS)]
        n = edges[i % len(edges)] if i % 3 == 0 else int(rs.randint(0, 65537))
        blocks.append(fuzzgen.gen(k, 9000 + i, n).tobytes())
    got = compress_many(blocks, varint=True, hw_compat=True)
    for b, g in zip(blocks, got):
        assert g == _varint(len(b)) + hw_model.compress_bytes(b), len(b)
        assert oracle_lib.uncompress(g) == b


def test_the_flag_is_refused_where_it_has_no_meaning():
    import torch
    P = pkg()
    src = torch.zeros(65536 * 4, dtype=torch.uint8, device="cuda")
    slots = torch.empty(4 * P.SLOT_STRIDE, dtype=torch.uint8, device="cuda")
    lens = torch.empty(4, dtype=torch.int32, device="cuda")
    rc = P.lib().sz_compress_stream_hip(src.data_ptr(), 65536 * 4, slots.data_ptr(), P.SLOT_STRIDE, lens.data_ptr(),
                                        P.FLAG_HW_COMPAT, None)
    assert rc == P.SZ_EINVAL
    rc = P.lib().sz_compress_blocks_hip(src.data_ptr(), 65536, None, 4, slots.data_ptr(), P.SLOT_STRIDE, lens.data_ptr(), 4, None)
    assert rc == P.SZ_EINVAL
