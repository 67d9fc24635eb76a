"""Executable model of the fixed-tile dense path of szk::compress_one_block (snappy_compress.hip,
"FAST DENSE PATH").  TEST INFRASTRUCTURE: it states, in plain Python integers, the mask algebra the
kernel performs with 64-bit ballots (walk, exit state, conflict cut, deferred emission), so that the
algebra can be checked against the oracle on the CPU (tests/test_tile_model.py) where there is no GPU.

The serial parts of the parse (first scan of a block, strided probes, block tail) are the plain
serial algorithm of SURVEY.md Appendix A here; the kernel has its own wave-parallel forms of those.

Lane l of tile t stands for position 64*t + l.  Masks are Python ints used as 64-bit words.
"""
M64 = (1 << 64) - 1
KMUL = 0x1E35A7BD


def ld32(d, p):
    return d[p] | (d[p + 1] << 8) | (d[p + 2] << 16) | (d[p + 3] << 24)


def _probe_offsets(n=448):
    off, skip, pos = [], 32, 0
    for _ in range(n):
        off.append(pos)
        step = skip >> 5
        skip += step
        pos += step
    return off


OFF = _probe_offsets()  # distance of the k-th probe of a scan from its first one (skip_k = 32 + OFF[k])


def below(k):  # bits 0..k-1, k in 0..64
    return (1 << k) - 1


def ctz(x):
    return (x & -x).bit_length() - 1


def top(x):  # index of the highest set bit
    return x.bit_length() - 1


class Model:
    def __init__(self, data, fast=True, margin=128, stats=None, sparse=False):
        self.sparse = sparse  # scans past their 32nd probe stay in the tile path (first segment of a visit)
        self.d = bytes(data)
        self.n = len(self.d)
        self.out = bytearray()
        self.fast = fast
        self.margin = margin
        self.stats = stats if stats is not None else {}

    # ---------------------------------------------------------------- element encoders (App. A)
    def emit_literal(self, start, ln):
        v = ln - 1
        if v < 60:
            self.out.append(v << 2)
        elif v < 256:
            self.out += bytes([60 << 2, v])
        else:
            self.out += bytes([61 << 2, v & 0xFF, v >> 8])
        self.out += self.d[start:start + ln]

    def copy64(self, off, ln):
        if ln < 12 and off < 2048:
            self.out += bytes([1 | ((ln - 4) << 2) | ((off >> 8) << 5), off & 0xFF])
        else:
            self.out += bytes([2 | ((ln - 1) << 2), off & 0xFF, off >> 8])

    def emit_copy(self, off, ln):
        while ln >= 68:
            self.copy64(off, 64)
            ln -= 64
        if ln > 64:
            self.copy64(off, 60)
            ln -= 60
        self.copy64(off, ln)

    def H(self, p):
        return ((ld32(self.d, p) * KMUL) & 0xFFFFFFFF) >> self.shift

    def match_len(self, cand, ip, m=4):
        d, n = self.d, self.n
        while ip + m < n and d[cand + m] == d[ip + m]:
            m += 1
        return m

    # ---------------------------------------------------------------- whole fragment
    def compress(self):
        n, d = self.n, self.d
        if n == 0:
            return bytes(self.out)
        T = 256
        while T < 16384 and T < n:
            T *= 2
        self.shift = 32 - (T.bit_length() - 1)
        self.tab = [0] * T
        self.next_emit = 0
        if n >= 15:
            self.ip_limit = n - 15
            self.serial_and_fast()
        if self.next_emit < n:
            self.emit_literal(self.next_emit, n - self.next_emit)
        return bytes(self.out)

    def serial_and_fast(self):
        """State: P = next position to probe; chain = P ends a copy; sc = probes done in this scan."""
        tab, d = self.tab, self.d
        P, chain, skip = 1, False, 32
        hold = False  # the fast path declined this state (a scan may reach probe 32): one serial step first
        while True:
            if (chain or self.sparse) and self.fast and not hold and P + self.margin <= self.ip_limit:
                P0, c0, s0 = P, chain, (skip - 32 if chain else None)
                P, chain, sc = self.fast_path(P, chain, 0 if chain else self.sc_of_skip(skip))
                skip = 32 + OFF[sc]
                hold = P == P0
                continue
            hold = False
            if chain:
                if P >= self.ip_limit:
                    return
                tab[self.H(P - 1)] = P - 1
                h = self.H(P)
                cand = tab[h]
                tab[h] = P
                if ld32(d, P) == ld32(d, cand):
                    m = self.match_len(cand, P)
                    self.emit_copy(P - cand, m)
                    P += m
                    self.next_emit = P
                    continue
                P += 1
                chain, skip = False, 32
                continue
            step = skip >> 5
            skip += step
            if P + step > self.ip_limit:
                return
            h = self.H(P)
            cand = tab[h]
            tab[h] = P
            if ld32(d, P) == ld32(d, cand):
                self.emit_literal(self.next_emit, P - self.next_emit)
                m = self.match_len(cand, P)
                self.emit_copy(P - cand, m)
                P += m
                self.next_emit = P
                chain = True
                continue
            P += step

    # ---------------------------------------------------------------- the fast dense path
    def stage(self, t):
        """Per-lane data of tile t against the table as it stands: hash, candidate, hit, local match
        length 4..32 (32 = "long": goes on past the local bytes)."""
        d = self.d
        h, cand, hit, ml = [0] * 64, [0] * 64, 0, [0] * 64
        for l in range(64):
            p = 64 * t + l
            h[l] = self.H(p)
            cand[l] = self.tab[h[l]]
        self.cmp_lanes(t, h, cand, ml, range(64))
        for l in range(64):
            if ld32(d, 64 * t + l) == ld32(d, cand[l]):
                hit |= 1 << l
        return h, cand, hit, ml

    def cmp_lanes(self, t, h, cand, ml, lanes):
        d = self.d
        for l in lanes:
            p = 64 * t + l
            m = 0
            while m < 32 and d[cand[l] + m] == d[p + m]:  # (fast path: p + 32 <= n always)
                m += 1
            ml[l] = m

    @staticmethod
    def sc_of_skip(skip):  # probe index of a scan whose skip counter stands at `skip`
        return OFF.index(skip - 32)

    def fast_path(self, P, chain=True, sc=0):
        """Entered with chain == True at P.  Returns (P, chain, sc) for the serial code, or
        (None, ..) never (the tail is always left to the serial code)."""
        st = self.stats
        d, tab = self.d, self.tab
        t, e = P >> 6, P & 63
        c_in, sc_in = chain, sc
        if chain:  # the reference inserts P-1 before it looks P up
            tab[self.H(P - 1)] = P - 1
        h, cand, Hraw, ml = self.stage(t)
        Vacc = Hpacc = 0
        acc_open, acc_emit = (not chain), self.next_emit
        while True:
            st["visits"] = st.get("visits", 0) + 1
            Hm = Hraw & ~below(e) & M64
            A = M64  # lanes of the first segment that are probed (sparse mode: a pattern past probe 32)
            A_scan, sc0, nxt_pos, nxt_k, h1 = 0, 0, 64, 0, None
            if not self.sparse:
                # ---- slow-path conditions: a scan could reach probe 32 inside this tile
                flagged = False
                for l in range(e, 33):
                    tl = Hm >> l
                    if tl == 0 or ctz(tl) >= 32:
                        flagged = True
                lead = (ctz(Hm) if Hm else 64) - e
                if not c_in and sc_in + lead > 31:
                    flagged = True
                if flagged:
                    st["flagged"] = st.get("flagged", 0) + 1
                    self.flush(t, Vacc, Hpacc, acc_open, acc_emit, cand, ml, 64)
                    return 64 * t + e, c_in, sc_in
            else:
                # ---- the probes of the first segment: the chain probe at e (c_in), then the scan's schedule
                sc0 = 0 if c_in else sc_in
                pos, k = (e + 1, 0) if c_in else (e, sc_in)
                if (32 + OFF[k]) >> 5 > 32:  # (a scan whose steps have grown past half a tile: the general code)
                    st["flagged"] = st.get("flagged", 0) + 1
                    self.flush(t, Vacc, Hpacc, acc_open, acc_emit, cand, ml, 64)
                    return 64 * t + e, c_in, sc_in
                while pos < 64:
                    A_scan |= 1 << pos
                    pos += (32 + OFF[k]) >> 5
                    k += 1
                nxt_pos, nxt_k = pos, k
                A = A_scan | ((1 << e) if c_in else 0)
                H1 = Hm & A
                h1 = ctz(H1) if H1 else None
            # ---- walk: vJ[l] = where the parse goes from lane l (next hit at or after l, then past its match)
            vJ = [64] * 64
            for l in range(64):
                tl = Hm >> l
                if tl:
                    nh = l + ctz(tl)
                    mw = 64 if ml[nh] == 32 else ml[nh]
                    vJ[l] = min(nh + mw, 64)
            lim = 64
            seg1 = 0  # lanes of the first segment up to and including its hit (sparse mode)
            if not self.sparse:
                S, jj = 0, e
                while jj < 64:
                    S |= 1 << jj
                    jj = vJ[jj]
            else:
                S = 1 << e
                if h1 is None:
                    seg1 = M64 & ~below(e)
                else:
                    seg1 = below(h1 + 1) & ~below(e)
                    jj = min(h1 + (64 if ml[h1] == 32 else ml[h1]), 64)
                    while jj < 64:
                        tl = Hm >> jj
                        if jj <= 32 and (tl == 0 or ctz(tl) >= 32):  # this segment would leave the dense regime inside the tile:
                            lim = jj                                  # the visit ends here; it becomes a first segment
                            st["far_cuts"] = st.get("far_cuts", 0) + 1
                            break
                        S |= 1 << jj
                        jj = vJ[jj]
                Hm = Hm & (~seg1 | ((1 << h1) if h1 is not None else 0)) & M64  # hits at lanes the scan skips do not exist
            # ---- exit state for limit `lim`, commit, conflict check; at most one repeat with lim = cut
            for attempt in range(2):
                bl = below(lim)
                Hm_t = Hm & bl
                Sx = S & bl
                s_l = top(Sx)
                Hx = Hm_t >> s_l
                Hp = Hm_t & ~((Hm_t - Sx) & M64) & M64
                long_lane = None
                new_emit = self.next_emit
                if Hx == 0:  # the literal / scan is still open at lim
                    bitlim = (1 << lim) if lim < 64 else 0
                    V = (((Hp << 1) | bitlim) - Sx) & M64 & bl
                    chain2, P2 = False, 64 * t + lim
                    if self.sparse and Sx == (1 << e):  # still in the first segment: its schedule says where it stands
                        rest = A_scan & ~bl & M64
                        P2 = 64 * t + (ctz(rest) if rest else nxt_pos)
                        sc2 = sc0 + bin(A_scan & bl).count("1")
                        if c_in:
                            new_emit = 64 * t + e
                    elif Sx == (1 << e) and not c_in:
                        sc2 = sc_in + lim - e
                    else:
                        new_emit = 64 * t + s_l
                        sc2 = lim - s_l - 1
                else:
                    V = (((Hp << 1) & M64) - Sx) & M64
                    jx = s_l + ctz(Hx)
                    assert jx == top(Hp)
                    if ml[jx] == 32:
                        long_lane = jx
                        chain2, P2, sc2 = True, None, 0
                    else:
                        P2 = 64 * t + jx + ml[jx]
                        chain2, sc2 = True, 0
                        new_emit = P2
                Ins = (Sx & ~(1 << e)) >> 1
                pm1_done = lim == 64 and chain2 and P2 == 64 * t + 64
                if pm1_done:  # the copy ends with the tile: its last byte is P-1 of the next tile's first probe
                    Ins |= 1 << 63
                assert V & Ins == 0
                C = (V & (A | ~seg1) & M64) | Ins  # (sparse mode: lanes the first segment's scan skips are not inserted)
                # conflict: two lanes of C share a bucket -> the later one should have seen the earlier
                seen, cut = {}, 64
                for l in range(64):
                    if (C >> l) & 1:
                        if h[l] in seen:
                            cut = min(cut, l)
                        else:
                            seen[h[l]] = l
                if cut == 64:
                    break
                assert attempt == 0 and cut > e
                st["cuts"] = st.get("cuts", 0) + 1
                lim = cut
            for l in range(64):  # commit (lanes of C have pairwise different buckets now)
                if (C >> l) & 1:
                    tab[h[l]] = 64 * t + l
            self.next_emit = new_emit
            Vacc |= V
            Hpacc |= Hp
            if long_lane is not None:
                self.flush(t, Vacc, Hpacc, acc_open, acc_emit, cand, ml, long_lane)
                ip = 64 * t + long_lane
                m = self.match_len(cand[long_lane], ip, 32)
                self.emit_copy(ip - cand[long_lane], m)
                P2 = ip + m
                self.next_emit = P2
                Vacc = Hpacc = 0
                acc_open, acc_emit = False, P2
                st["long"] = st.get("long", 0) + 1
            t2, e2 = P2 >> 6, P2 & 63
            if t2 != t:
                if Vacc:
                    self.flush(t, Vacc, Hpacc, acc_open, acc_emit, cand, ml, 64)
                Vacc = Hpacc = 0
                acc_open, acc_emit = (not chain2), self.next_emit
                if P2 + self.margin > self.ip_limit:
                    return P2, chain2, sc2
                if chain2 and not pm1_done:
                    tab[self.H(P2 - 1)] = P2 - 1  # (pm1_done: went in as Ins bit 63 of the tile before)
                t = t2
                h, cand, Hraw, ml = self.stage(t)
                st["tiles"] = st.get("tiles", 0) + 1
            else:
                # same tile again (after a cut or a long match): insert P-1, then refresh the lookups of
                # the lanes whose bucket was written since they were made
                if chain2:
                    tab[h[e2 - 1]] = 64 * t + e2 - 1
                stale = [l for l in range(e2, 64) if tab[h[l]] != cand[l]]
                if stale:
                    st["patches"] = st.get("patches", 0) + 1
                    for l in stale:
                        cand[l] = tab[h[l]]
                    self.cmp_lanes(t, h, cand, ml, stale)
                    for l in stale:
                        if ld32(d, 64 * t + l) == ld32(d, cand[l]):
                            Hraw |= 1 << l
                        else:
                            Hraw &= ~(1 << l)
            e, c_in, sc_in = e2, chain2, sc2

    def flush(self, t, Vacc, Hpacc, acc_open, acc_emit, cand, ml, long_lane):
        """Elements closed inside the accumulated lanes of tile t: literal runs that end at a hit of
        the path, and the copies of those hits (the long one is emitted by the caller)."""
        if not Hpacc:
            return
        jfirst, jlast = ctz(Hpacc), top(Hpacc)
        Lit = Vacc & ~Hpacc & below(jlast)
        if acc_open:
            self.emit_literal(acc_emit, 64 * t + jfirst - acc_emit)
            Lit &= ~below(jfirst)
        RS = Lit & ~(Lit << 1)
        for l in range(64):
            p = 64 * t + l
            if (RS >> l) & 1:
                runlen = ctz(Hpacc >> l)
                self.emit_literal(p, runlen)
            elif (Hpacc >> l) & 1 and l != long_lane:
                self.copy64(p - cand[l], ml[l])


def compress_fragment(data, fast=True, stats=None, sparse=False):
    return Model(data, fast=fast, stats=stats, sparse=sparse).compress()
