#!/usr/bin/env python3
"""Instruction-count study of EXACT 11 x 11 medians of 8-bit images on gfx950 ("evaluate on paper /
in a CPU script first ... go to the GPU only if the count is <= ~90 lane-instructions per pixel").  CPU only.

It prices, in one currency, the select the product runs and the work-sharing alternatives:

  A  the product: bit-sliced radix select (d2pc_median_bs_tile.hpp), counted from the structure of its loop
  B  sorted columns + pruned merge networks (the verdict's proposal), built HERE as explicit comparator networks over the 121
     window taps (Batcher odd-even merges, comparators that cannot reach the median removed by backward liveness, one-sided
     comparators priced as such), with everything that neighbouring windows can share counted once:
       B1  merge tree: pairs P(x), quads Q(x), octets O(x) of sorted columns, each computed once per column position
       B2  Young tableau: sorted columns, sorted rows, corner pruning (only 73 of 121 taps can be the median), select among them
     priced (a) on packed 16-bit lanes (v_pk_min_u16 / v_pk_max_u16: two pixels per instruction, but these issue every FOUR
     cycles: profiles/r02_valu_rates.txt) and (b) bit-sliced (a compare-exchange of two 8-bit numbers held as 8 plane words of 32
     pixels = 8 v_bitop3 for the borrow chain + 16 for the two selects = 24 two-cycle instructions per 32 pixels)
  C  the radix select with SHARED column counts for the top L bit planes (fixed thresholds: a column's count of taps below
     64 / 128 / 192 ... serves every window that holds the column)

Currency: 2-cycle VALU issue slots per pixel ("lane-instructions per pixel" x (cycles / 2)): what a SIMD-32 spends.
"""
import functools
import sys

K = 11
N = K * K
RANK = N // 2  # 0-based rank of the median


# ---------------------------------------------------------------- A: the product's select
def product_select():
    """per thread (32 pixels) and tile: planes 7..0; per plane 121 AND + carry-save tree over 121 + 7 words + finish + 7 rank
    muxes + 121 candidate updates (none in the last plane)."""
    def csa_ops(n_inputs):
        # a full adder = 2 bitop3 per 3 words of one weight -> 1 sum + 1 carry: words drop by 1 per FA at a level
        ops, level = 0, n_inputs
        weights = [level]
        w = 0
        while w < len(weights):
            n = weights[w]
            fa = 0
            while n >= 3:
                n -= 2
                fa += 1
            ops += 2 * fa
            if fa:
                if w + 1 == len(weights):
                    weights.append(0)
                weights[w + 1] += fa
            weights[w] = n
            w += 1
        # finish: ripple over the levels, <= 2 ops per level
        ops += 2 * len(weights)
        return ops
    per_plane = N + csa_ops(N + 7) + 7 + 1
    total = 8 * per_plane + 7 * N
    return total, per_plane


def product_select_peeled():
    """since round 6 (D2PC_BS_PEEL_MSB): in the most significant plane every tap is a candidate -- no AND in its count, its update is the
    XOR that creates the candidate words, and the k*k moves that set them to all-ones (not counted in product_select) are gone"""
    total, per_plane = product_select()
    return total - N, total + N


# ---------------------------------------------------------------- B: comparator networks
def oddeven_merge(lo_wires, hi_wires):
    """Batcher's odd-even merge of two sorted wire lists (any lengths): comparators as (a, b) = (min wire, max wire)."""
    a, b = list(lo_wires), list(hi_wires)
    if not a or not b:
        return []
    if len(a) == 1 and len(b) == 1:
        return [(a[0], b[0])]
    # merge evens with evens, odds with odds, then fix neighbours of the interleaved result
    ev = oddeven_merge(a[0::2], b[0::2])
    od = oddeven_merge(a[1::2], b[1::2])
    net = ev + od
    # after the two sub-merges the sequences E (evens merged) and O (odds merged) live on these wires, in this order:
    E = sorted_positions(a[0::2], b[0::2])
    O = sorted_positions(a[1::2], b[1::2])
    # result order: E0, then pairs (O_i, E_{i+1}) compared, ...
    for i in range(min(len(O), len(E) - 1)):
        net.append((O[i], E[i + 1]))
    return net


def sorted_positions(a, b):
    """wires that hold the merged sequence of a-merge-b, in ascending order, for the odd-even construction below"""
    return merged_order(tuple(a), tuple(b))


@functools.lru_cache(maxsize=None)
def merged_order(a, b):
    """order of wires after oddeven_merge(a, b): (the recursion mirrors oddeven_merge)"""
    if not a:
        return list(b)
    if not b:
        return list(a)
    if len(a) == 1 and len(b) == 1:
        return [a[0], b[0]]
    E = merged_order(a[0::2], b[0::2])
    O = merged_order(a[1::2], b[1::2])
    out = [E[0]]
    i = 0
    while i < len(O) or i + 1 < len(E):
        if i < len(O) and i + 1 < len(E):
            out += [O[i], E[i + 1]]
        elif i < len(O):
            out.append(O[i])
        else:
            out.append(E[i + 1])
        i += 1
    return out


def apply(net, vals):
    v = list(vals)
    for a, b in net:
        if v[a] > v[b]:
            v[a], v[b] = v[b], v[a]
    return v


def check_merge():
    import random
    rng = random.Random(1)
    for la, lb in ((11, 11), (22, 22), (44, 44), (22, 11), (88, 33), (5, 9), (1, 7)):
        a = list(range(la))
        b = list(range(la, la + lb))
        net = oddeven_merge(a, b)
        order = merged_order(tuple(a), tuple(b))
        for _ in range(60):
            va = sorted(rng.randrange(256) for _ in range(la))
            vb = sorted(rng.randrange(256) for _ in range(lb))
            out = apply(net, va + vb)
            got = [out[w] for w in order]
            assert got == sorted(va + vb), (la, lb)
    return True


def prune(net, live_out):
    """backward liveness: a comparator whose two outputs are dead is removed; one live output = a min-only or max-only
    comparator.  -> (full, half)"""
    live = set(live_out)
    full = half = 0
    for a, b in reversed(net):
        la, lb = a in live, b in live
        if not (la or lb):
            continue
        if la and lb:
            full += 1
        else:
            half += 1
        live.add(a)
        live.add(b)
    return full, half


def merge_tree():
    """B1.  Wires: column c holds taps 11 c .. 11 c + 10, sorted (the column sort is priced separately)."""
    col = [list(range(11 * c, 11 * c + 11)) for c in range(11)]
    levels = []
    # shared nodes, each computed ONCE per column position and used by every window that contains it:
    P = oddeven_merge(col[0], col[1])                                     # pairs   P(x) = L(x) + L(x+1)
    pw = merged_order(tuple(col[0]), tuple(col[1]))
    P2 = oddeven_merge(col[2], col[3])
    pw2 = merged_order(tuple(col[2]), tuple(col[3]))
    Q = oddeven_merge(pw, pw2)                                            # quads   Q(x) = P(x) + P(x+2)
    qw = merged_order(tuple(pw), tuple(pw2))
    # octet O(x) = Q(x) + Q(x+4): build the second quad on columns 4..7
    pw3 = merged_order(tuple(col[4]), tuple(col[5]))
    pw4 = merged_order(tuple(col[6]), tuple(col[7]))
    qw2 = merged_order(tuple(pw3), tuple(pw4))
    O = oddeven_merge(qw, qw2)
    ow = merged_order(tuple(qw), tuple(qw2))
    # of the 88 sorted taps of an octet only ranks 27..60 can be the median of 121 (r smaller inside, 87 - r larger inside)
    keep = [ow[r] for r in range(88) if r <= RANK and 87 - r <= RANK]
    o_full, o_half = prune(O, keep)
    # R(x) = P(x) + L(x+2): the window's last three columns
    pw5 = merged_order(tuple(col[8]), tuple(col[9]))
    R = oddeven_merge(pw5, col[10])
    rw = merged_order(tuple(pw5), tuple(col[10]))
    # final: rank 60 of (kept octet ranks 27..60, i.e. 34 taps with 27 known-smaller taps dropped) + R (33 taps): the median
    # is the tap of rank 60 - 27 = 33 of those 67
    F = oddeven_merge(keep, rw)
    fw = merged_order(tuple(keep), tuple(rw))
    f_full, f_half = prune(F, [fw[RANK - 27]])
    levels.append(("pairs  P(x)  merge(11,11), once per column", len(P), 0))
    levels.append(("quads  Q(x)  merge(22,22), once per column", len(Q), 0))
    levels.append(("octets O(x)  merge(44,44) pruned to ranks 27..60", o_full, o_half))
    levels.append(("R(x) = P + L merge(22,11), once per column", len(R), 0))
    levels.append(("final: rank 33 of 34 + 33 taps, pruned to ONE output", f_full, f_half))
    return levels


def batcher_sort(wires):
    if len(wires) <= 1:
        return [], list(wires)
    h = len(wires) // 2
    na, oa = batcher_sort(wires[:h])
    nb, ob = batcher_sort(wires[h:])
    return na + nb + oddeven_merge(oa, ob), merged_order(tuple(oa), tuple(ob))


SORT11 = 35   # comparators of the best known 11-input sorting network (Knuth, TAOCP 3, 5.3.4)
SORT10, SORT9, SORT8 = 29, 25, 19


def young_tableau():
    """B2.  After sorting the columns and then the rows of the 11 x 11 tap matrix, tap (a, b) (1-based sorted positions) has
    >= a b - 1 taps below it and >= (12 - a)(12 - b) - 1 above: only taps with a b <= 61 and (12 - a)(12 - b) <= 61 can be the
    median.  Rows are per-window work (a row of the column-sorted matrix mixes 11 image columns): 11 sorts of 11, of which a
    row only has to deliver its surviving positions (partial sort: priced by liveness on Batcher's network)."""
    band = [(a, b) for a in range(1, 12) for b in range(1, 12) if a * b <= 61 and (12 - a) * (12 - b) <= 61]
    below = sum(1 for a in range(1, 12) for b in range(1, 12) if (12 - a) * (12 - b) > 61)
    rows_full = rows_half = 0
    for a in range(1, 12):
        wires = list(range(11))
        net, order = batcher_sort(wires)
        keep = [order[b - 1] for (aa, b) in band if aa == a]
        f, h = prune(net, keep)
        rows_full += f
        rows_half += h
    # select rank (60 - below) among the band's taps: a pruned Batcher sort of them (their partial order is ignored: an UPPER
    # bound of this step; exploiting it is what the diagonal passes of Adams 2021 do)
    wires = list(range(len(band)))
    net, order = batcher_sort(wires)
    f, h = prune(net, [order[RANK - below]])
    return len(band), below, (rows_full, rows_half), (f, h)


def column_sort_shared():
    """Sorted column of 11 rows per (x, y): two vertically adjacent outputs share 10 rows -- sort those once (29), insert the
    11th tap per output (a bubble of 10 comparators): (29 + 2 x 10) / 2 per output; every sorted column serves the 11 windows
    that contain it, i.e. ONE column sort per output pixel (times the tile's halo)."""
    return (SORT10 + 2 * 10) / 2.0


# ---------------------------------------------------------------- C: shared column counts for the top L planes
def shared_counts(L, halo):
    """radix select whose first L planes use column counts at the 2^L - 1 fixed thresholds (shared by the windows holding the
    column); per thread (32 pixels).  -> (VALU ops for those L planes incl. the candidate masks the later planes need, ops the
    product spends on the same planes)"""
    thr = 2 ** L - 1
    pre = halo * (thr * (1 + 16) + 11 * 0)           # per (word column, row): B_T from L plane words + 11-tap count, per threshold
    window = 0
    for l in range(1, L + 1):
        window += (2 ** (l - 1) - 1) * 4 * K         # pick each column's 4-bit count for the window's own prefix
        window += 74 + 12                            # add 11 four-bit numbers (37 full adders) + rank test and update
    cand_init = N * L                                # the candidate words of plane L+1: tap's top L bits == the median's
    _, per_plane = product_select()
    return pre + window + cand_init, L * (per_plane + N)


def main():
    assert check_merge()
    tot, per_plane = product_select()
    px = 32.0
    print("# Exact 11 x 11 median of 8-bit pixels on gfx950: VALU cost per pixel in 2-cycle issue slots (a wave64 instruction of the")
    print("# 2-cycle class -- v_and, v_xor, v_bitop3 -- is one slot per lane; the 4-cycle class -- v_pk_min/max_u16, v_max_u32 -- two)")
    print(f"A  product, bit-sliced radix select: {per_plane} instructions per plane and 32 pixels, {tot} per 32 pixels"
          f" = {tot / px:.1f} slots / pixel   [measured: 300.9 M VALU wave-instructions per 16 x 4K launch = 154 / pixel incl. the other stages]")
    peeled, before = product_select_peeled()
    print(f"   with the {N} moves that initialise the candidate words: {before} = {before / px:.1f}; since the most significant plane is peeled off (round 6): "
          f"{peeled} = {peeled / px:.1f} slots / pixel   [measured: 284.9 M per launch = 146 / pixel]")
    cs = column_sort_shared()
    print(f"\nB  sorted columns + pruned merges.  Column sort, shared vertically in pairs: {cs:.1f} comparators per pixel (x halo)")
    ce_pk = 2 * 2 / 2.0      # min + max, 4-cycle class (2 slots each), two pixels per instruction
    ce_bs, half_bs = 24 / px, 16 / px
    half_pk = 1 * 2 / 2.0
    print(f"   price of a comparator: packed u16 {ce_pk:.2f} slots / pixel (one-sided {half_pk:.2f}); bit-sliced {ce_bs:.3f} (one-sided {half_bs:.3f})")
    levels = merge_tree()
    full = sum(f for _, f, _ in levels)
    half = sum(h for _, _, h in levels)
    print("   B1 merge tree (every shared node once per column position):")
    for name, f, h in levels:
        print(f"        {name:58s} {f:4d} two-sided + {h:3d} one-sided")
    b1_pk = (cs + full) * ce_pk + half * half_pk
    b1_bs = (cs + full) * ce_bs + half * half_bs
    print(f"      total {cs + full:.0f} + {half} comparators per pixel -> packed u16 {b1_pk:.0f} slots / pixel, bit-sliced {b1_bs:.0f}")
    nb, below, (rf, rh), (sf, sh) = young_tableau()
    print(f"   B2 Young tableau: {nb} of 121 taps can be the median after column + row sorts ({below} known smaller);")
    print(f"        11 row sorts delivering only those positions: {rf} two-sided + {rh} one-sided; select among the {nb}: {sf} + {sh}")
    b2_pk = (cs + rf + sf) * ce_pk + (rh + sh) * half_pk
    b2_bs = (cs + rf + sf) * ce_bs + (rh + sh) * half_bs
    print(f"      total {cs + rf + sf:.0f} + {rh + sh} comparators per pixel -> packed u16 {b2_pk:.0f} slots / pixel, bit-sliced {b2_bs:.0f}")
    print(f"   (the verdict's gate: <= ~90 per pixel.  Packed 16-bit lanes miss it by {min(b1_pk, b2_pk) / 90:.1f} x; bit-sliced comparators tie the")
    print("    product's count at best -- and need 8 registers per tap (a sorted column = 88 VGPRs, a window's 11 columns = 968), i.e.")
    print("    LDS round trips of 16 words in and 16 out around every 24-instruction comparator)")
    print("\nC  radix select with column counts shared between windows for the top L planes (halo = word columns per thread column:")
    print("   18 / 8 in the product's 256-wide tile; 1.0 = an infinitely wide tile)")
    for halo in (2.25, 1.0):
        for L in (1, 2, 3, 4):
            new, old = shared_counts(L, halo)
            print(f"      halo {halo:4.2f}  L = {L}: {new:6.0f} instructions per 32 pixels instead of {old:5d}: {100.0 * (old - new) / tot:5.1f} % of the select")
    return 0


if __name__ == "__main__":
    sys.exit(main())
