#!/usr/bin/env python3
"""How much can ANY hand-out of k_fused's 16-row blocks to the eight waves of a workgroup shorten the aggregation phase?
For every graph of the C3 batch (500 ER N=200 p=0.1): trips of each block (rows in descending entry count, 16 per block, 4
entries per trip - fused.hip: row_blocks_init), the load of the slowest wave under (a) the static deal the kernel uses
(block k*8 + w to wave w, the second pass reversed), (b) the best hand-out of whole blocks (longest-processing-time first =
what a ticket counter over blocks in descending order does, idealised: no cost per hand-out), (c) perfectly divisible work
(sum / 8).  CPU only; python tools/deal_bound.py [graphs] [nodes] [p]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from distgcn_amd import datagen
B = int(sys.argv[1]) if len(sys.argv) > 1 else 500
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
p = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
waves = 8
hb = datagen.er_batch(B, n, p)
stat, lpt, ideal, w0 = [], [], [], []
for n0, n1 in hb.graph_slices():
    cnt = np.diff(hb.row_ptr[n0:n1 + 1]) + 1              # entries of L's rows (diagonal first)
    order = np.sort(cnt)[::-1]
    blocks = [max(1, (int(order[i]) + 3) // 4) for i in range(0, len(order), 16)]   # trips = those of the block's first row
    load = np.zeros(waves)
    for k, t in enumerate(blocks):
        pas, j = divmod(k, waves)
        load[(waves - 1 - j) if (pas & 1) else j] += t
    stat.append(load.max()); w0.append(load[0])
    l2 = np.zeros(waves)
    for t in blocks:                                       # descending already: LPT
        l2[l2.argmin()] += t
    lpt.append(l2.max())
    ideal.append(sum(blocks) / waves)
stat, lpt, ideal, w0 = map(np.array, (stat, lpt, ideal, w0))
print("%d ER N=%d p=%g graphs, %d waves: trips of the slowest wave per layer, mean over graphs" % (B, n, p, waves))
print("  static deal (the kernel's)          %.2f   (wave 0: %.2f -> waits %.0f %% of the phase)" % (stat.mean(), w0.mean(), 100 * (1 - w0.mean() / stat.mean())))
print("  best hand-out of whole blocks (LPT) %.2f   (%.1f %% shorter; graphs where it is shorter at all: %d of %d)" % (lpt.mean(), 100 * (1 - lpt.mean() / stat.mean()), int((lpt < stat).sum()), B))
print("  perfectly divisible (sum / waves)   %.2f   (%.1f %% shorter)" % (ideal.mean(), 100 * (1 - ideal.mean() / stat.mean())))
