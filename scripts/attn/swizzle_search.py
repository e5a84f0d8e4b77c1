"""Search an XOR swizzle X(row) (on the 16-byte unit index of a plane row) that makes BOTH LDS access patterns of the bf16-plane attention
kernels (csrc/rt_attention_v2.hip) bank-conflict free on gfx950: ds_read_b128 operand reads (rows = lanes) and ds_read_b64_tr_b16
transpose reads (4-row blocks).  Bank model: MI355X_MICROARCH.md §LDS (64 banks x 4 B; b128 in four 16-lane groups, tr_b16 in two
32-lane halves).  Layout: row r, plane p at r * 3 * ROWB + p * ROWB; X is linear over GF(2) in the row's low 5 bits."""
import itertools
import sys

HD = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ROWB = HD * 2
UNITS = HD // 8                      # 16-byte units per plane row
UB = UNITS.bit_length() - 1          # bits of the unit index
B128_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
               list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def X(mat, r):
    v = 0
    for b in range(UB):
        bit = 0
        for rb in range(5):
            if (mat[b] >> rb) & 1 and (r >> rb) & 1:
                bit ^= 1
        v |= bit << b
    return v


def extra_cycles(addrs_bytes, width):
    """extra LDS cycles of one lane group: max over banks of distinct addresses on that bank, minus 1"""
    per_bank = {}
    for a in addrs_bytes:
        for w in range(width // 4):
            per_bank.setdefault(((a // 4) + w) % 64, set()).add(a)
    return max(len(v) for v in per_bank.values()) - 1


def cost(mat):
    tot = 0
    for R0 in (0, 16):                       # b128 operand reads: lane (i, g) reads row R0 + i, unit 4s + g
        for s in range(HD // 32):
            for p in range(3):
                for grp in B128_GROUPS:
                    addrs = []
                    for l in grp:
                        i, g = l & 15, l >> 4
                        r = R0 + i
                        addrs.append(r * 3 * ROWB + p * ROWB + (((4 * s + g) % UNITS) ^ X(mat, r)) * 16)
                    tot += extra_cycles(addrs, 16)
    for kb in range(2):                      # transpose reads: lane (g, j, t) supplies row 16kb + 4g + j, 8-byte chunk 4cb + t
        for cb in range(HD // 16):
            for half in range(2):
                addrs = []
                for l in range(32 * half, 32 * half + 32):
                    g, i = l >> 4, l & 15
                    j, t = i >> 2, i & 3
                    k = 16 * kb + 4 * g + j
                    c = 4 * cb + t
                    addrs.append(k * 3 * ROWB + ((c ^ (X(mat, k) << 1)) * 8))
                tot += extra_cycles(addrs, 8)
    return tot


best = None
for mat in itertools.product(range(32), repeat=UB):
    c = cost(mat)
    if best is None or c < best[0]:
        best = (c, mat)
        print(c, [bin(m) for m in mat], flush=True)
        if c == 0:
            break
print("best", best, "identity-like (r>>1)&7 cost", cost(tuple(2 << b for b in range(UB))), "no swizzle", cost((0,) * UB))
