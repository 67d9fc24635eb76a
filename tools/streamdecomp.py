#!/usr/bin/env python3
"""tools/streamdecomp.py -- device-side rate of decoding ONE raw stream resident in HBM
(sz_uncompress_stream_hip: fragment index + one wave per fragment) against the single-wave decoder.
usage: streamdecomp.py [MiB]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import fuzzgen
from pkg_loader import load_package
P = load_package()
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for name, corpus in (("text", fuzzgen.corpus("mtg")), ("random", fuzzgen.corpus("random_ascii")), ("noise", None)):
    total = mib << 20
    if corpus is None:  # every fragment one literal: the index's pass 0 covers the stream
        data = np.random.default_rng(7).integers(0, 256, total, dtype=np.uint8).tobytes()
    else:
        data = (corpus * (total // len(corpus) + 1))[:total]
    comp = P.compress(data)
    n = len(comp)
    body = 1
    while comp[body - 1] & 0x80:
        body += 1
    buf = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
    buf[:n] = torch.from_numpy(np.frombuffer(comp, dtype=np.uint8).copy()).cuda()
    dst = torch.empty(total, dtype=torch.uint8, device="cuda")
    for _ in range(2):
        _, st = P.uncompress_stream(buf, n, body, total, dst=dst)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); _, st = P.uncompress_stream(buf, n, body, total, dst=dst); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ok = st.cpu().numpy().tolist() == [0, 0, 0] and dst.cpu().numpy().tobytes() == data
    line = "%s: one stream, %d MiB out, %d bytes in: indexed decode %.3f ms = %.1f GB/s of output (%s)" % (
        name, mib, n, min(ts), total / min(ts) / 1e6, "ok" if ok else "MISMATCH")
    if mib <= 64:
        ln = torch.tensor([n], dtype=torch.int32, device="cuda")
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); P.uncompress_blocks(buf, n, ln, 1, dst_stride=total, dst=dst); b.record(); torch.cuda.synchronize()
        line += "; single wave %.1f ms" % a.elapsed_time(b)
    print(line, flush=True)
