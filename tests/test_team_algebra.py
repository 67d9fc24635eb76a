"""The team kernel's serial algebra (snappy_compress.hip, "TEAM FAST DENSE PATH") in plain Python integers, checked
on the CPU against the single-wave forms it replaces (tests/tile_model.py holds those against the oracle):

  * the per-lane jump table X packs, beside where the parse goes from a lane, the exit state of a visit whose last
    jump starts there -- the walk's last v_readlane IS the exit state -- and must equal exit_for(64) of the fixed-tile
    path for every hit mask, match lengths, entry lane and entry state;
  * a visit that ends before a cut (two inserted lanes share a bucket) takes its exit from the same table;
  * the shared-bucket test by masks made ahead of the visit (G(lane) = earlier lanes of the tile with the same hash)
    finds the same cut as "the second inserted lane of a bucket";
  * look-ups made ahead of the token, re-read when it arrives, with only the changed lanes redone, equal fresh ones.
TEST INFRASTRUCTURE (no GPU)."""
import random

M64 = (1 << 64) - 1


def ctz(x):
    return (x & -x).bit_length() - 1


def top(x):
    return x.bit_length() - 1


def below(k):
    return (1 << k) - 1


def exit_for(Hm, S, lim, e, c_in, sc_in, next_emit, tmlw, base):
    """The fixed-tile path's exit computation (snappy_compress.hip `exit_for`, tests/tile_model.py)."""
    bl = below(lim)
    Hm_t, Sx = Hm & bl, S & bl
    s_l = top(Sx)
    Hx = Hm_t >> s_l
    open_ = 1 if Hx == 0 else 0
    Hp = Hm_t & ~((Hm_t - Sx) & M64) & M64
    bitlim = (1 << lim) if lim < 64 else 0
    V = ((((Hp << 1) & M64) | (bitlim if open_ else 0)) - Sx) & M64 & bl
    jx = (s_l + ctz(Hx | (1 << 63))) & 63
    mj = tmlw[jx]
    carried = (1 if Sx == (1 << e) else 0) & (c_in ^ 1)
    is_long = (open_ ^ 1) & (1 if mj == 64 else 0)
    pend = base + jx + mj
    P2 = base + lim if open_ else (base + jx if is_long else pend)
    sc2 = (sc_in + lim - e if carried else lim - s_l - 1) if open_ else 0
    emit2 = (next_emit if carried else base + s_l) if open_ else pend
    pm1 = (open_ ^ 1) & (is_long ^ 1) & (1 if lim == 64 else 0) & (1 if jx + mj == 64 else 0)
    Ins = ((Sx & ~(1 << e)) >> 1) | (pm1 << 63)
    return dict(P2=P2, chain2=open_ ^ 1, sc2=sc2, emit2=(None if is_long else emit2), pm1=pm1,
                ll=jx if is_long else 64, V=V, Hp=Hp, C=V | Ins)


def jumps(Hraw, tmlw, base):
    """X, Y, Z of every lane (the kernel's `jumps` lambda): exits point at themselves with bit 6 set."""
    X, Y, Z = [0] * 64, [0] * 64, [0] * 64
    for l in range(64):
        tl = Hraw >> l
        if tl == 0:
            X[l] = l | 64 | (1 << 7) | (64 << 16)
            Y[l] = (base + 64) | ((63 - l) << 19)
            Z[l] = base + l
            continue
        nh = l + ctz(tl)
        m = tmlw[nh]
        lng = 1 if m == 64 else 0
        s = nh + m
        tgt = min(max(s, l + 1), 64)
        p2off = nh if lng else s
        X[l] = (l | 64 if tgt >= 64 else tgt) | (lng << 8) | (nh << 10) | ((p2off & 0xff) << 16)
        Y[l] = (base + p2off) | (1 << 17) | ((1 if (not lng and s == 64) else 0) << 18)
        Z[l] = base + p2off
    return X, Y, Z


def walk(X, e, steps_per_group=4):
    """The branch-free walk: groups of `steps_per_group` steps, exits loop onto themselves."""
    S, x = 1 << e, X[e]
    while True:
        for _ in range(steps_per_group):
            S |= 1 << (x & 63)
            x = X[x & 63]
        if x & 64:
            return S, x


def team_exit(Hm, X, Y, Z, e, c_in, sc_in, next_emit):
    S, x = walk(X, e)
    s_l = x & 63
    y, z = Y[s_l], Z[s_l]
    open_ = (x >> 7) & 1
    carried = (1 if s_l == e else 0) & (c_in ^ 1) & open_
    is_long = (x >> 8) & 1
    pm1 = (y >> 18) & 1
    Hp = Hm & ~((Hm - S) & M64) & M64
    V = (((Hp << 1) & M64) - S) & M64
    C = V | ((S & ~(1 << e)) >> 1) | (pm1 << 63)
    return dict(P2=y & 0x1ffff, chain2=(y >> 17) & 1, sc2=(sc_in + 64 - e) if carried else (y >> 19) & 63,
                emit2=None if is_long else (next_emit if carried else z), pm1=pm1,
                ll=((x >> 10) & 63) if is_long else 64, V=V, Hp=Hp, C=C), S


def team_exit_cut(Hm, X, Y, Z, S, cut, e, c_in, sc_in, next_emit, base):
    bl = below(cut)
    Sx, Hm_t = S & bl, Hm & bl
    sl = top(Sx)
    xs = X[sl]
    nh = (xs >> 10) & 63
    open_ = ((xs >> 7) & 1) | (1 if nh >= cut else 0)
    carried = (1 if sl == e else 0) & (c_in ^ 1) & open_
    Hp = Hm_t & ~((Hm_t - Sx) & M64) & M64
    V = ((((Hp << 1) & M64) | ((1 << cut) if open_ else 0)) - Sx) & M64 & bl
    is_long = (open_ ^ 1) & ((xs >> 8) & 1)
    return dict(P2=base + cut if open_ else Y[sl] & 0x1ffff, chain2=open_ ^ 1, pm1=0,
                sc2=((sc_in + cut - e) if carried else cut - sl - 1) if open_ else 0,
                emit2=None if is_long else ((next_emit if carried else base + sl) if open_ else Z[sl]),
                ll=nh if is_long else 64, V=V, Hp=Hp, C=V | ((Sx & ~(1 << e)) >> 1))


def _random_tile(rng):
    dens = rng.choice([0.02, 0.1, 0.3, 0.6, 0.9])
    Hraw = 0
    for l in range(64):
        if rng.random() < dens:
            Hraw |= 1 << l
    tmlw = [rng.choice([4, 4, 5, 6, 8, 11, 12, 20, 31, 64]) for _ in range(64)]
    return Hraw, tmlw


def test_the_walks_last_readlane_is_the_exit_state():
    rng = random.Random(1)
    for _ in range(30000):
        Hraw, tmlw = _random_tile(rng)
        e, c_in, sc_in = rng.randrange(64), rng.randrange(2), rng.randrange(32)
        base = 64 * rng.randrange(2, 1000)
        next_emit = base - rng.randrange(1, 30)
        Hm = Hraw & ~below(e) & M64
        X, Y, Z = jumps(Hraw, tmlw, base)
        new, S = team_exit(Hm, X, Y, Z, e, c_in, sc_in, next_emit)
        # the starts of the serial walk: from each start to the next hit and past its match
        S2, j = 0, e
        while j < 64:
            S2 |= 1 << j
            tl = Hm >> j
            j = 64 if not tl else min(max(j + ctz(tl) + tmlw[j + ctz(tl)], j + 1), 64)
        assert S == S2
        assert new == exit_for(Hm, S2, 64, e, c_in, sc_in, next_emit, tmlw, base)


def test_a_visit_cut_short_takes_its_exit_from_the_same_table():
    rng = random.Random(2)
    for _ in range(30000):
        Hraw, tmlw = _random_tile(rng)
        e, c_in, sc_in = rng.randrange(63), rng.randrange(2), rng.randrange(32)
        base = 64 * rng.randrange(2, 1000)
        next_emit = base - rng.randrange(1, 30)
        Hm = Hraw & ~below(e) & M64
        X, Y, Z = jumps(Hraw, tmlw, base)
        _, S = team_exit(Hm, X, Y, Z, e, c_in, sc_in, next_emit)
        cut = rng.randrange(e + 1, 64)
        assert team_exit_cut(Hm, X, Y, Z, S, cut, e, c_in, sc_in, next_emit, base) == \
            exit_for(Hm, S, cut, e, c_in, sc_in, next_emit, tmlw, base)


def test_shared_buckets_known_ahead_give_the_same_cut():
    """G(lane) = earlier lanes of the tile with the lane's hash; conflict lanes = lanes of C whose G meets C; the
    lowest one is the cut: the second inserted lane of the first bucket that has two (the read-back method's)."""
    rng = random.Random(3)
    for _ in range(20000):
        nb = rng.choice([40, 200, 2000])  # buckets in play: few -> many shared ones
        h = [rng.randrange(nb) for _ in range(64)]
        C = rng.getrandbits(64)
        G = [sum(1 << j for j in range(l) if h[j] == h[l]) for l in range(64)]
        conf = sum(1 << l for l in range(64) if (C >> l) & 1 and (G[l] & C))
        seen, cut = {}, 64
        for l in range(64):
            if (C >> l) & 1:
                if h[l] in seen:
                    cut = min(cut, l)
                else:
                    seen[h[l]] = l
        assert (ctz(conf) if conf else 64) == cut
        # the revisit: a lane whose bucket this turn has written takes the last writer below it as its candidate
        for l in range(64):
            m = G[l] & C
            want = max((j for j in range(l) if (C >> j) & 1 and h[j] == h[l]), default=None)
            assert (top(m) if m else None) == want


def test_lookups_made_ahead_and_rechecked_equal_fresh_ones():
    """The stale rule: candidates read from the table at some earlier time, re-read when the token arrives; only
    the lanes whose entry changed are redone.  Entries only ever move forward, so `changed` <=> `stale`."""
    rng = random.Random(4)
    for _ in range(5000):
        nb = 64
        table = [rng.randrange(0, 1000) for _ in range(nb)]
        h = [rng.randrange(nb) for _ in range(64)]
        ahead = [table[x] for x in h]  # look-ups made ahead of the token
        for _ in range(rng.randrange(0, 40)):  # the turns in between insert later positions
            table[rng.randrange(nb)] = rng.randrange(1000, 2000)
        now = [table[x] for x in h]
        stale = [l for l in range(64) if now[l] != ahead[l]]
        merged = list(ahead)
        for l in stale:
            merged[l] = now[l]
        assert merged == now
