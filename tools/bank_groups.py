#!/usr/bin/env python3
"""Expected LDS cycles per 16-lane bank group of the aggregation's ds_read_b128 ("what bounds it").

A group holds four row slots; each reads one 64-byte half (= 16 banks = one quarter of the 64) of a gathered Z1 row, and
the complementary half in the next instruction.  Which quarter a half row sits in is a property of the image layout; a
group costs as many cycles as its busiest quarter has readers.  Enumerates all combinations of four gathered rows for:
the layout of the kernel (quarter = 2 * row parity + half, halves fixed per slot), the same with the best per-entry choice
of which half goes first, and placements in which the rows of a 4-row period use the quarter pairs given."""
import itertools


def cycles(rows, hs):
    cnt = [0] * 4
    for r, h in zip(rows, hs):
        cnt[r[h]] += 1
    return max(cnt)


def expect(pairs, fixed=None):
    tot = n = 0
    for rows in itertools.product(pairs, repeat=4):
        if fixed is not None:
            c = cycles(rows, fixed) + cycles(rows, tuple(1 - h for h in fixed))
        else:
            c = min(cycles(rows, hs) + cycles(rows, tuple(1 - h for h in hs)) for hs in itertools.product((0, 1), repeat=4))
        tot += c
        n += 1
    return tot / n / 2


parity = [(0, 1), (2, 3)]
cyclic = [(0, 1), (2, 3), (1, 2), (3, 0)]
print("kernel layout, halves fixed per slot (0, 1, 0, 1):   %.3f cycles per group" % expect(parity, (0, 1, 0, 1)))
print("kernel layout, best half order per entry:            %.3f" % expect(parity))
print("cyclic quarter pairs, halves fixed:                  %.3f" % expect(cyclic, (0, 1, 0, 1)))
print("cyclic quarter pairs, best half order per entry:     %.3f" % expect(cyclic))
