"""Lane-level replay of `topk_coarse_frag_kernel` (scripts/wip/topk_fragment_major.patch) in numpy: the fragment-major images as
`rt_one_plane_to_fragments` writes them, the workgroup / wave / block-pair walk with its prefetch cursor, the operand each lane hands to
v_mfma_f32_32x32x16_bf16 and the accumulator element it reads back — against plain bf16 dot products.  Checks the INDEX arithmetic of the
kernel (which unit a lane loads, which item row and user an accumulator register belongs to, which blocks a workgroup visits, the odd last
pair), not the hardware.  python scripts/wip/emu_topk_frag.py"""
import numpy as np

IB = 128           # catalog rows per item block (rt_topk.hip)
P = 8              # fragment loads in flight per wave and item block
IW = 2


def bf16_rne(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32).view(np.float32)


def one_plane_units(x):
    """[n, d] fp32 -> [n, d / 8, 8] bf16 values (as fp32): unit u of a row = its k = 8 u .. 8 u + 7."""
    h = bf16_rne(x)
    return h.reshape(x.shape[0], x.shape[1] // 8, 8)


def to_fragments(units, rows_pad):
    """rt_one_plane_to_fragments: unit (r, u) -> unit ((r / 32) n_s + u / 2) 64 + 32 (u & 1) + r % 32; rows up to rows_pad zero."""
    n, n_units, _ = units.shape
    n_s = n_units // 2
    out = np.zeros((rows_pad // 32 * n_s * 64, 8), dtype=np.float32)
    for r in range(n):
        for u in range(n_units):
            out[((r // 32) * n_s + u // 2) * 64 + 32 * (u & 1) + r % 32] = units[r, u]
    return out


def mfma_32x32x16(a_lanes, b_lanes, acc):
    """acc[lane][reg] += sum over the k = 16 slot; A lane = 32 kb + i holds row i's k block kb (8 values), B lane = 32 kb + j column j's.
    Accumulator register r of lane (half, col): row (r & 3) + 8 (r >> 2) + 4 half, column col."""
    a = a_lanes.reshape(2, 32, 8)          # [kb][i][t]
    b = b_lanes.reshape(2, 32, 8)          # [kb][j][t]
    full = np.einsum("kit,kjt->ij", a.astype(np.float64), b.astype(np.float64))     # [row i][col j]
    for lane in range(64):
        half, col = lane >> 5, lane & 31
        for r in range(16):
            row = (r & 3) + 8 * (r >> 2) + 4 * half
            acc[lane, r] += full[row, col]


def run_workgroup(items_frag, users_frag, n_s, TU, sx, S, blk_begin, blk_end, user0, scores_out, n_cand, n_users):
    UB = 32 * TU
    n_blocks = blk_end - blk_begin
    my_blocks = (n_blocks - sx + S - 1) // S if (sx < S and n_blocks > sx) else 0
    if my_blocks == 0:
        return
    ufrag = users_frag[(user0 >> 5) * n_s * 64:(user0 >> 5) * n_s * 64 + TU * n_s * 64]       # the LDS copy
    last_blk = blk_begin + sx + (my_blocks - 1) * S
    n_pairs = (my_blocks + IW - 1) // IW
    for wave in range(4):
        def frag0(blk):
            return ((blk * (IB // 32) + wave) * n_s) * 64           # + lane

        # issue cursor
        state = {"is": 0, "ij": 0, "ip": None}

        def set_pair(j):
            ip = []
            for iw in range(IW):
                blk = min(blk_begin + sx + (j * IW + iw) * S, last_blk)
                ip.append(frag0(blk))
            state["ip"] = ip

        def issue():
            got = [items_frag[state["ip"][iw]:state["ip"][iw] + 64].copy() for iw in range(IW)]     # 64 lanes x 8 values
            for iw in range(IW):
                state["ip"][iw] += 64
            state["is"] += 1
            if state["is"] == n_s:
                state["is"] = 0
                state["ij"] += 1
                set_pair(state["ij"] if state["ij"] < n_pairs else n_pairs - 1)
            return got

        set_pair(0)
        abuf = [issue() for _ in range(P)]
        bf = [None, None]
        bf[0] = [ufrag[(tu * n_s) * 64:(tu * n_s) * 64 + 64] for tu in range(TU)]
        for j in range(n_pairs):
            acc = np.zeros((IW, TU, 64, 16))
            for s0 in range(0, n_s, P):
                for q in range(P):
                    nxt = 0 if s0 + q + 1 == n_s else s0 + q + 1
                    bf[(q + 1) & 1] = [ufrag[(tu * n_s + nxt) * 64:(tu * n_s + nxt) * 64 + 64] for tu in range(TU)]
                    af = abuf[q]
                    abuf[q] = issue()
                    for iw in range(IW):
                        for tu in range(TU):
                            mfma_32x32x16(af[iw], bf[q & 1][tu], acc[iw, tu])
            for iw in range(IW):
                blk = blk_begin + sx + (j * IW + iw) * S
                if blk > last_blk:
                    continue
                pos0 = blk * IB
                for tu in range(TU):                       # select_block's view of the accumulators
                    for lane in range(64):
                        half, col = lane >> 5, lane & 31
                        u = user0 + tu * 32 + col
                        for r in range(16):
                            row = (r & 3) + 8 * (r >> 2) + 4 * half
                            p = pos0 + wave * 32 + row
                            if u < n_users and p < n_cand:
                                assert np.isnan(scores_out[u, p]), "pair scored twice"
                                scores_out[u, p] = acc[iw, tu, lane, r]


def main():
    rng = np.random.default_rng(0)
    for (n_cand, d, n_users, TU, S, phases) in ((128 * 5 + 77, 128, 150, 4, 2, 1), (128 * 7, 256, 40, 2, 3, 2), (300, 128, 33, 1, 1, 1)):
        UB = 32 * TU
        items = rng.normal(size=(n_cand, d)).astype(np.float32)
        users = rng.normal(size=(n_users, d)).astype(np.float32)
        n_s = d // 16
        assert n_s % P == 0
        rows_pad = (n_cand + 127) // 128 * 128
        users_pad = (n_users + UB - 1) // UB * UB
        users_pad = (users_pad + 127) // 128 * 128
        items_frag = to_fragments(one_plane_units(items), rows_pad)
        users_frag = to_fragments(one_plane_units(users), users_pad)
        n_blocks = (n_cand + IB - 1) // IB
        n_tiles = (n_users + UB - 1) // UB
        scores = np.full((n_users, n_cand), np.nan)
        bounds = np.linspace(0, n_blocks, phases + 1).astype(int)           # phases: seed / resume launches over block ranges
        for b0, b1 in zip(bounds[:-1], bounds[1:]):
            for ty in range(n_tiles):
                for sx in range(S):
                    run_workgroup(items_frag, users_frag, n_s, TU, sx, S, int(b0), int(b1), ty * UB, scores, n_cand, n_users)
        assert not np.isnan(scores).any(), "a (user, item) pair was never scored"
        ref = bf16_rne(users).astype(np.float64) @ bf16_rne(items).astype(np.float64).T
        err = np.abs(scores - ref).max()
        print(f"n_cand {n_cand} d {d} users {n_users} TU {TU} S {S} phases {phases}: every pair scored once, max |emu - ref| = {err:.2e}")
        assert err < 1e-9


if __name__ == "__main__":
    main()
