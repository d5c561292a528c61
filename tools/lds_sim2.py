#!/usr/bin/env python3
"""LDS model of the interval-scheme wave kernel (variant 8); see lds_sim.py for the rules."""
import sys, re, collections
import numpy as np
sys.path.insert(0, __file__.rsplit('/', 2)[0]); sys.path.insert(0, __file__.rsplit('/', 1)[0])
from lds_sim import cost
from oracle import oracle as O

def interval_tables(n_mels=80):
    w = O.mel_filterbank(16000, 400, n_mels)[:, :200]
    edges = O.mel_frequencies(n_mels + 2, 0.0, 8000.0)
    idx = [min(n_mels, int(np.sum(edges[1:n_mels + 1] <= 40.0 * k))) for k in range(200)]
    n_int = n_mels + 1
    first = [0] * n_int; cnt = [0] * n_int
    for k, i in enumerate(idx):
        if cnt[i] == 0: first[i] = k
        cnt[i] += 1
    ns = (n_int + 10) // 11
    lens, starts = [], []
    for s in range(ns):
        L = max(cnt[i] for i in range(s * 11, min(n_int, s * 11 + 12)))
        lens.append(L)
        row = []
        for j in range(12):
            i = s * 11 + j
            st = first[i] if i < n_int and cnt[i] else 0
            if st + L > 200: st = 200 - L
            row.append(st)
        starts.append(row)
    return starts, lens

def simulate(XS=436, XR=20, PS=201, TW1S=44, starts=None, lens=None, perm=None):
    cat = collections.Counter(); catc = collections.Counter()
    def acc(kind, addrs, tag):
        c, x = cost(kind, addrs); cat[tag] += c; catc[tag] += x
    lanes = [(l, l // 11, l % 11) for l in range(55)]
    lanes3 = [(l, l // 12, l % 12) for l in range(60)]
    BL = 100000
    P = (lambda k: perm[k]) if perm is not None else (lambda k: k)
    for n1 in range(20): acc('r64', {l: BL + 20 * n1 + 2 * j for l, f, j in lanes if j < 10}, 'win')
    for k1 in range(1, 20): acc('r64', {l: BL + 400 + j * TW1S + 2 * k1 for l, f, j in lanes if j < 10}, 'tw1')
    for k1 in range(21): acc('w64', {l: f * XS + k1 * XR + 2 * j for l, f, j in lanes if j < 10}, 'xw')
    for i in range(5):
        acc('r128', {l: f * XS + j * XR + 4 * i for l, f, j in lanes}, 'u')
        acc('r128', {l: f * XS + (20 if j == 0 else 20 - j) * XR + 4 * i for l, f, j in lanes}, 'v')
        acc('r128', {l: BL + 860 + j * 20 + 4 * i for l, f, j in lanes}, 'tw2')
    for q in range(10):
        acc('w32', {l: f * PS + P(j + 20 * q) for l, f, j in lanes}, 'pk')
        acc('w32', {l: f * PS + P(200 - j - 20 * q) for l, f, j in lanes}, 'pm')
    for i, L in enumerate(lens):
        for r in range(L):
            acc('r32', {l: f * PS + P(starts[i][j] + r) for l, f, j in lanes3}, 'p3')
            acc('r64', {l: BL + 2000 + (r * 12 + j) * 2 for l, f, j in lanes3}, 'w3')
    return sum(cat.values()), sum(catc.values()), dict(cat), dict(catc)

if __name__ == '__main__':
    starts, lens = interval_tables(80)
    print(simulate(starts=starts, lens=lens))
    res = []
    for XS in range(420, 452, 4):
        for PS in range(201, 233, 2):
            t, c, _, _ = simulate(XS, 20, PS, 44, starts, lens)
            res.append((t, c, XS, PS))
    res.sort(); print(res[:8])
