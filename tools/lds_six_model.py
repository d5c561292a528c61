#!/usr/bin/env python3
"""LDS bank model of whisper400_six_runs_kernel (csrc/whisper_six.hpp; 6 frames x 10 lanes per wave), the rules of tools/lds_sim.py.
Prints the LDS-array cycles of one unit per access class and the conflict cycles among them; `search` anneals the exchange-row order
(SixLayout::row_pos) and tries the power-row pitches (SixLayout::kPStride).   tools/lds_six_model.py [search [seconds]]"""
import sys, random, time
import numpy as np
sys.path.insert(0, __file__.rsplit('/', 2)[0]); sys.path.insert(0, __file__.rsplit('/', 1)[0])
from lds_sim import cost
from oracle import oracle as O

ROWPOS = [14, 1, 13, 4, 16, 3, 17, 11, 6, 8, 19, 9, 7, 15, 18, 5, 2, 12, 0, 10]
LENS = [1, 1, 1, 2, 2, 3, 4, 6, 7]


def interval_starts(n_mels=80, lanes=10):
    w = O.mel_filterbank(16000, 400, n_mels)[:, :200]
    idx = [-1] * 200
    for k in range(200):
        nz = np.nonzero(w[:, k])[0]
        if len(nz) == 0: continue
        lo, hi = int(nz[0]), int(nz[-1])
        if hi != lo: idx[k] = hi
        else:
            peak = int(np.argmax(w[lo]))
            idx[k] = lo if k <= peak else lo + 1
    n_int = n_mels + 1
    first = [0] * n_int; cnt = [0] * n_int
    for k, i in enumerate(idx):
        if i < 0: continue
        if cnt[i] == 0: first[i] = k
        cnt[i] += 1
    real = lanes - 1
    ns = (n_int + real - 1) // real
    starts, lens = [], []
    for s in range(ns):
        L = max(cnt[i] for i in range(s * real, min(n_int, s * real + lanes)))
        lens.append(L)
        row = []
        for j in range(lanes):
            i = s * real + j
            st = first[i] if i < n_int and cnt[i] else 0
            if st + L > 200: st = 200 - L
            row.append(st)
        starts.append(row)
    return starts, lens


def simulate(rp=ROWPOS, XS=404, XR=20, PS=213, starts=None, lens=None, parts=None):
    cat = {}
    def acc(kind, addrs, tag):
        if parts is not None and tag not in parts: return
        c, x = cost(kind, addrs)
        t = cat.setdefault(tag, [0, 0]); t[0] += c; t[1] += x
    lanes = [(l, l // 10, l % 10) for l in range(60)]
    BL = 1 << 20
    for n1 in range(0, 20, 2): acc('r128', {l: BL + 44 * j + 2 * n1 for l, f, j in lanes}, 'win')
    for k1 in range(1, 20): acc('r64', {l: BL + 440 + 44 * j + 2 * k1 for l, f, j in lanes}, 'tw1')
    for k1 in range(20): acc('w64', {l: f * XS + rp[k1] * XR + 2 * j for l, f, j in lanes}, 'xw')
    for i in range(5):
        acc('r128', {l: f * XS + rp[j] * XR + 4 * i for l, f, j in lanes}, 'u')
        acc('r128', {l: f * XS + rp[10 if j == 0 else 20 - j] * XR + 4 * i for l, f, j in lanes}, 'v')
        acc('r128', {l: BL + 880 + 28 * j + 4 * i for l, f, j in lanes}, 'tw2')
    for s in range(0, 10, 2):
        def kk(j): return (j if s < 6 else (-110 if j == 0 else j)) + 20 * s
        acc('w32', {l: f * PS + kk(j) for l, f, j in lanes}, 'pk')
        acc('w32', {l: f * PS + kk(j) + 20 for l, f, j in lanes}, 'pk')
        acc('w32', {l: f * PS + 200 - kk(j) for l, f, j in lanes}, 'pm')
        acc('w32', {l: f * PS + 180 - kk(j) for l, f, j in lanes}, 'pm')
    for i, L in enumerate(lens):
        for r in range(L):
            acc('r32', {l: f * PS + starts[i][j] + r for l, f, j in lanes}, 'p3')
            acc('r64', {l: BL + 4096 + 2 * j + 20 * r for l, f, j in lanes}, 'w3')
    acc('w32', {l: 1280 + 12 * f + j for l, f, j in lanes}, 'pmaxw')
    for q in range(2): acc('r128', {l: 1280 + 12 * f + 4 * q for l, f, j in lanes if j < 9}, 'pmaxr')
    acc('r64', {l: 1280 + 12 * f + 8 for l, f, j in lanes if j < 9}, 'pmaxr')
    tot = sum(v[0] for v in cat.values()); conf = sum(v[1] for v in cat.values())
    return tot, conf, cat


if __name__ == '__main__':
    starts, lens = interval_starts()
    assert lens == LENS, lens
    t, c, cat = simulate(starts=starts, lens=lens)
    print("shipped: LDS cycles per unit", t, "conflicts", c, "(%.3f)" % (c / t), {k: tuple(v) for k, v in cat.items()})
    if len(sys.argv) > 1 and sys.argv[1] == 'search':
        budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
        rng = random.Random(7)
        best = None
        for PS in [p for p in range(201, 240) if (6 * p) <= 6 * 404]:
            _, cp, _ = simulate(ROWPOS, 404, 20, PS, starts, lens, parts={'pk', 'pm', 'p3'})
            if best is None or cp < best[0]: best = (cp, PS)
            print("PS", PS, "power-row conflicts", cp)
        print("best power pitch", best)
        # exchange-row order: anneal on the u / v conflicts (the writes do not depend on it)
        def uv(rp): return simulate(rp, 404, 20, 213, starts, lens, parts={'u', 'v'})[1]
        cur = list(ROWPOS); cc = uv(cur); bestrp = (cc, list(cur))
        t0 = time.time(); T = 2.0
        while time.time() - t0 < budget:
            a, b = rng.sample(range(20), 2)
            cur[a], cur[b] = cur[b], cur[a]
            nc = uv(cur)
            if nc <= cc or rng.random() < pow(2.718, -(nc - cc) / T):
                cc = nc
                if cc < bestrp[0]: bestrp = (cc, list(cur)); print("u+v conflicts", cc, cur, flush=True)
                if cc == 0: break
            else:
                cur[a], cur[b] = cur[b], cur[a]
            T = max(0.05, T * 0.9995)
        print("best row order", bestrp)
