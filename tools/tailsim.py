#!/usr/bin/env python3
"""tools/tailsim.py -- how a launch of the team kernel ends, as an event-driven simulation on the CPU (no GPU needed):
1 024 workgroups, each a team (a block in TT ms) and two lone waves (TL ms), one queue; policies for the end of the
launch: none (round 5), intra (a team with nothing left adopts the block of a lone wave of its own workgroup at a team's
pace: what runs since round 6), cross (... of any workgroup), +sib / +stop (lone waves that do not take one of the last
`thr` blocks when their sibling already has / at all).  Constant rates, so optimistic about what adoption gains (on the
device the lone waves run faster while teams are idle): measured 16.07 -> 15.79 ms at 16 384 blocks where this says
15.79 -> 15.20.  What it is for: the ORDER of the policies (EXPERIMENTS.md R6.6): the board across workgroups adds 1 %,
the hand-out rules nothing.
usage: python tools/tailsim.py"""
import heapq, random, sys
def sim(nblocks, policy, TT=1.75, TL=3.85, nwg=1024, jitter=0.03, seed=1, adopt_min=0.25, thr=0, copy=0.02):
    """policy: none | intra | cross | intra+sib | cross+sib.  returns end time.
    Work of a block = 1.0; team speed 1/TT, lone speed 1/TL."""
    rnd = random.Random(seed)
    nextb = 0
    # worker state
    ev = []  # (time, kind, wg, idx)
    lone = {}  # (wg,i) -> dict(start, dur, done(bool), adopted)
    late = {}  # wg -> count of lone waves in a 'late' block
    for wg in range(nwg):
        heapq.heappush(ev, (rnd.random()*0.02, 'team_free', wg, 0))
        for i in range(2):
            heapq.heappush(ev, (rnd.random()*0.02, 'lone_free', wg, i))
    end = 0.0
    pool = []  # cross: list of adoptable lone blocks (wg,i)
    idle_teams = []
    def lone_remaining(key, t):
        L = lone[key]
        return max(0.0, 1.0 - (t - L['start'])/L['dur'])  # fraction of block left
    def try_adopt(wg, t):
        # returns True if the team of wg adopted something
        cands = []
        if policy.startswith('intra'):
            cands = [(wg, i) for i in range(2)]
        elif policy.startswith('cross'):
            cands = [(wg, i) for i in range(2)] + pool_list()
        best = None
        for key in cands:
            L = lone.get(key)
            if not L or L['gone']: continue
            rem = lone_remaining(key, t)
            if rem * L['dur'] < adopt_min: continue
            if best is None or rem > best[0]: best = (rem, key)
        if best is None: return False
        rem, key = best
        lone[key]['gone'] = True
        d = copy + rem * TT * (1 + rnd.uniform(-jitter, jitter))
        heapq.heappush(ev, (t + d, 'team_free', wg, 0))
        return True
    def pool_list():
        return [k for k, L in lone.items() if not L['gone']]
    while ev:
        t, kind, wg, i = heapq.heappop(ev)
        if kind == 'team_free':
            end = max(end, t)
            if nextb < nblocks:
                nextb += 1
                heapq.heappush(ev, (t + TT*(1+rnd.uniform(-jitter, jitter)), 'team_free', wg, 0))
            else:
                if policy != 'none':
                    try_adopt(wg, t)
        elif kind == 'lone_free':
            key = (wg, i)
            L = lone.get(key)
            if L is not None:
                if L['gone'] and not L.get('finished'):  # was adopted: this event is stale
                    L['finished'] = True
                    continue
                L['gone'] = True; L['finished'] = True
            end = max(end, t)
            if nextb < nblocks:
                take = True
                left = nblocks - nextb
                if policy.endswith('+sib') and left < thr:
                    sib = lone.get((wg, 1-i))
                    if sib and not sib['gone'] and sib.get('late'):
                        take = False
                if policy.endswith('+stop') and left < thr:
                    take = False
                if take:
                    nextb += 1
                    d = TL*(1+rnd.uniform(-jitter, jitter))
                    lone[key] = dict(start=t, dur=d, gone=False, late=(nblocks - nextb) < thr)
                    heapq.heappush(ev, (t + d, 'lone_free', wg, i))
    return end
if __name__ == '__main__':
    for nb in (4096, 16384, 65536):
        base = sim(nb, 'none')
        print(nb, 'none %.2f' % base, ' '.join('%s %.2f' % (p, sim(nb, p)) for p in ('intra', 'cross')),
              ' '.join('%s/%d %.2f' % (p, th, sim(nb, p, thr=th)) for p in ('intra+sib', 'cross+sib', 'intra+stop') for th in (1024, 2048, 3072)))
