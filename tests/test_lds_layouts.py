"""LDS layouts of the LDS-DMA ring GEMM and of the fused kernel's re-order buffers (csrc/kernels_nn.hip: hsw / asw), checked against the bank
model of MI355X_MICROARCH.md (ds_read_b128: four groups of 16 lanes over a 256-byte window of 16-byte slots) — no GPU needed.

The DMA writes lane l's 16 bytes at piece_base + 16 l, so the swizzle is applied on the GLOBAL side: LDS position p of row r holds global chunk
p ^ sw(r); the reader, who wants chunk c, looks at position c ^ sw(r).  These tests pin (1) that the two sides agree, (2) that every fragment
read of the kernels is conflict-free, (3) that the un-swizzled layouts are not (i.e. the test would notice a regression)."""
import itertools

G128 = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
G128 = G128 + [[l + 32 for l in g] for g in G128]


def extra_cycles(addr_of_lane):
    """LDS cycles beyond the conflict-free four of one ds_read_b128 (byte address per lane)."""
    tot = 0
    for grp in G128:
        slots = {}
        for l in grp:
            a = addr_of_lane(l)
            assert a % 16 == 0
            slots.setdefault((a // 16) % 16, set()).add(a)
        tot += sum(len(v) - 1 for v in slots.values())
    return tot


def _device_fn(name):
    """The swizzle function AS WRITTEN in csrc/kernels_nn.hip: `__device__ __forceinline__ int <name>(int row) { return <expr>; }` — the
    expression is C integer arithmetic that is also valid Python, so the test evaluates the kernel's own source, not a copy of it."""
    import os
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "backscrub_amd", "csrc", "kernels_nn.hip")).read()
    m = re.search(r"__device__\s+__forceinline__\s+int\s+%s\(int row\)\s*\{\s*return\s+([^;]+);\s*\}" % name, src)
    assert m, "%s(int row) not found in kernels_nn.hip" % name
    expr = m.group(1)
    assert re.fullmatch(r"[\s\d()row&|^<>+\-]+", expr), "unexpected tokens in %s: %r" % (name, expr)
    return eval("lambda row: " + expr)                      # noqa: S307 — the character class above admits integer operators only


asw = _device_fn("asw")      # f32 activation rows of 128 bytes (8 chunks of 4 floats)


def test_the_kernels_use_the_swizzles_where_the_model_says():
    """the address expressions of the fused expand kernel's staged (LDS-DMA) operand buffer, as written in the source (float units there: 32 floats = 128 bytes per
    row, chunk << 2 = 16 bytes).  The ring GEMM whose weight rows used the second swizzle (hsw) was deleted in round 6."""
    import os
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "backscrub_amd", "csrc", "kernels_nn.hip")).read()
    for needle in ("li * 32 + (((2 * g) ^ asw(li)) << 2)", "4 * (lp ^ asw(8 * h + lr))"):
        assert needle in src, needle
    assert "pw_gemm_ring_k" not in src and "hsw(" not in src


def test_activation_rows_are_conflict_free_with_asw_and_not_without():
    for h in (0, 1):          # the two quads of a lane's 8 floats: chunks 2g and 2g + 1
        read = lambda sw: extra_cycles(lambda l: (l & 15) * 128 + (((2 * (l >> 4) + h) ^ sw(l & 15)) << 4))
        assert read(asw) == 0
        assert read(lambda r: 0) > 0


def test_dma_side_and_reader_side_agree():
    # A piece: 8 rows x 8 chunk positions, lane = (row l >> 3, position l & 7) loads global chunk position ^ asw(row)
    for piece in range(4):
        lds = {}
        for l in range(64):
            row = 8 * piece + (l >> 3)
            lds[(row, l & 7)] = (l & 7) ^ asw(row & 15)
        for row, c in itertools.product(range(8 * piece, 8 * piece + 8), range(8)):
            assert lds[(row, c ^ asw(row & 15))] == c


def test_fragment_reads_use_xor_16_for_the_second_quad():
    # kernels read the second quad of a lane's 8 floats at (address ^ 16): positions (2g) ^ s and (2g + 1) ^ s differ in bit 0 only
    for li, g in itertools.product(range(16), range(4)):
        a0 = li * 128 + (((2 * g) ^ asw(li)) << 4)
        a1 = li * 128 + (((2 * g + 1) ^ asw(li)) << 4)
        assert a1 == a0 ^ 16


# ---- the segment kernels' tiles (csrc/kernels_seg.hip): VERDICT r4 #4(i) — SQ counters showed 55 % / 41 % / 40 % of the LDS-active cycles of seg_k2_k / seg_head_k /
# seg_tail_k as bank-conflict cycles.  The model below prices every LDS access pattern of those kernels under the layouts they use now (swz_a / swz_b / swz_l, evaluated
# from the kernel source) and under the dense [pixel][16] layout of round 4.
W128 = [list(range(8 * k, 8 * k + 8)) for k in range(8)]        # ds_write_b128: eight groups of eight consecutive lanes, banks modulo 32 (a 128-byte window of eight slots)


def _cycles(groups, nslots, addr_of_lane, active=lambda l: True):
    """LDS-array cycles of one wave instruction: per lane group the largest number of DISTINCT addresses on one 16-byte slot (N-way = N x; equal addresses broadcast)"""
    tot = 0
    for grp in groups:
        slots = {}
        for l in grp:
            if active(l):
                a = addr_of_lane(l)
                assert a % 16 == 0
                slots.setdefault((a // 16) % nslots, set()).add(a)
        tot += max([len(v) for v in slots.values()] + [1])
    return tot


def rd128(f, active=lambda l: True):
    return _cycles(G128, 16, f, active)          # conflict-free: 4


def wr128(f, active=lambda l: True):
    return _cycles(W128, 8, f, active)           # conflict-free: 8 (the instruction itself costs 13: conflicts hurt beyond that)


def _seg_fn(name, args="int x"):
    import os
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "backscrub_amd", "csrc", "kernels_seg.hip")).read()
    m = re.search(r"__device__\s+__forceinline__\s+int\s+%s\(%s\)\s*\{\s*return\s+([^;]+);\s*\}" % (name, re.escape(args)), src)
    assert m, "%s(%s) not found in kernels_seg.hip" % (name, args)
    expr = m.group(1)
    assert re.fullmatch(r"[\s\d()xqhalf_swzbkSegLoStride&|^<>+\-*,]+", expr), "unexpected tokens in %s: %r" % (name, expr)
    return expr


swz_a = eval("lambda x: " + _seg_fn("swz_a"))                      # noqa: S307
swz_b = eval("lambda x: " + _seg_fn("swz_b"))                      # noqa: S307
swz_l = eval("lambda x: " + _seg_fn("swz_l"))                      # noqa: S307
kSegLoStride = 16
col_a = eval("lambda x, q: " + _seg_fn("col_a", "int x, int q"))                      # noqa: S307
col_b = eval("lambda x, q, half: " + _seg_fn("col_b", "int x, int q, int half"))      # noqa: S307
col_l = eval("lambda x, q: " + _seg_fn("col_l", "int x, int q"))                      # noqa: S307
mfma_lane = lambda l: (l & 15, l >> 4)       # (pixel li, quad g): MFMA operand / epilogue lanes, and the depthwise lanes of k3 / tail since round 5
dw_lane = lambda l: (l >> 2, l & 3)          # (pixel, quad): the depthwise lanes of head / k2


def test_the_stride_of_the_staged_window_is_the_one_the_model_uses():
    import os
    import re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "backscrub_amd", "csrc", "segments.hpp")).read()
    assert int(re.search(r"constexpr int kSegLoStride = (\d+);", hdr).group(1)) == kSegLoStride


def test_round4_dense_tiles_conflict_as_the_counters_said():
    dense = lambda x, q: x * 16 + 4 * q
    assert wr128(lambda l: 4 * dense(*mfma_lane(l))) == 32                                        # MFMA epilogue store: 4-way
    assert rd128(lambda l: 4 * dense(*mfma_lane(l))) == 8                                         # MFMA operand read: 2-way
    assert all(rd128(lambda l: 4 * dense(2 * dw_lane(l)[0] + fx, dw_lane(l)[1])) == 8 for fx in range(3))      # stride-2 depthwise taps: 2-way
    assert all(rd128(lambda l: 4 * ((((mfma_lane(l)[0] + c) // 2) * 20) + 4 * mfma_lane(l)[1])) == 8 for c in range(4))   # up-sampling taps at the padded stride 20: 2-way


def test_stride1_tiles_are_conflict_free():
    """k2's B tile, k3 / tail's z tile: MFMA epilogue store, MFMA operand read, stride-1 depthwise taps on the MFMA's lanes (k3 / tail), for every 16-column block"""
    for base in (0, 16):
        assert wr128(lambda l: 4 * col_a(base + mfma_lane(l)[0], mfma_lane(l)[1])) == 8
        assert rd128(lambda l: 4 * col_a(base + mfma_lane(l)[0], mfma_lane(l)[1])) == 4
    for fx in range(3):      # lanes 14, 15 repeat lane 13's columns (TC <= 14)
        assert rd128(lambda l: 4 * col_a(min(mfma_lane(l)[0], 13) + fx, mfma_lane(l)[1])) == 4
    # and the lane constants the kernels hoist: the swizzle of column 16 ct + li is that of li
    assert all(swz_a(16 * ct + li) == swz_a(li) and swz_b(16 * ct + li) == swz_b(li) for ct in range(4) for li in range(16))
    for half in (8, 14, 15, 16):
        assert all(col_b(16 * ct + li, q, half) == 8 * ct * 16 + col_b(li, q, half) for ct in range(3) for li in range(16) for q in range(4))


def test_deinterleaved_tiles_are_conflict_free_and_a_bijection():
    """k2's x ([BR][RW], RW a multiple of 16; odd widths are covered for the head's [AR][AC] tile, where the layout was measured 4-5 % slower and is not used — profiles/r05b):
    MFMA epilogue store + the nine taps of the stride-2 depthwise"""
    for width in (32, 16, 29, 27, 31):
        half = (width + 1) // 2
        cells = {col_b(x, q, half) for x in range(width) for q in range(4)}
        assert len(cells) == 4 * width and min(cells) == 0 and max(cells) == 16 * width - 4                 # every quad of every column has its own 16 bytes of the row
        for row in range(3):
            rb = row * width * 16
            for ct in range((width + 15) // 16):
                assert wr128(lambda l: 4 * (rb + col_b(16 * ct + mfma_lane(l)[0], mfma_lane(l)[1], half)), lambda l: 16 * ct + mfma_lane(l)[0] < width) == 8
            for fy in range(3):
                for fx in range(3):
                    assert rd128(lambda l: 4 * ((2 * row + fy) * width * 16 + col_b(2 * dw_lane(l)[0] + fx, dw_lane(l)[1], half)), lambda l: 2 * dw_lane(l)[0] + 2 < width) == 4


def test_staged_window_is_conflict_free_for_every_alignment():
    """k3 / tail's low-resolution window [LR][LC][16]: the cooperative store (lane = 4 * pixel + quad) and the four bilinear taps (two neighbouring lanes share a source
    pixel at 2x up-sampling) for every start column of the tile and every window width"""
    assert len({col_l(x, q) for x in range(16) for q in range(4)}) == 64
    for LC in range(5, 17):
        for r in range(3):
            rb = r * LC * kSegLoStride
            assert wr128(lambda l: 4 * (rb + col_l(l >> 2, l & 3)), lambda l: l < 4 * LC) == 8
            for c in range(8):
                assert rd128(lambda l: 4 * (rb + col_l((mfma_lane(l)[0] + c) // 2, mfma_lane(l)[1]))) == 4
