"""The graph-specialised program (csrc/gen_mid.cpp + mid_prelude.hip, compiled by hipRTC when a context is created): what can be checked
without a GPU — the source is generated for every shipped architecture, compiles for gfx950, needs no scratch memory for the headline graph (bounded for the two larger ones) and fits the 128-register budget of
a 1024-lane workgroup; hipRTC and the on-disk cache work with no device present."""
import os
import re
import subprocess

import pytest

from conftest import MODEL_KEYS, ROOT, model_path

HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def api():
    from backscrub_amd import api as a
    a.lib()
    return a


# scratch bytes tolerated per architecture: NONE since round 6.  Rounds 3-5 tolerated 512 / 1088 bytes for segm_full / MLKit: their kernels sat
# at 126-128 registers because the compiler kept every lane-derived value alive across the whole straight-line kernel; build_mid_kernel (bsx_api.hip) now compiles the
# form in which each op re-derives its lane indices wherever the plain form spills, and takes it if it spills less (MLKit: 352 -> 0 bytes, 128 -> 92 registers).
SCRATCH_LIMIT = {"lite": 0, "full": 0, "mlkit": 0}


@pytest.mark.parametrize("key", ["lite", "full", "mlkit"])
def test_generated_kernel_compiles_without_spills(api, key, tmp_path):
    src = api.model_kernel_source(model_path(key))
    assert 'extern "C" __global__' in src and "bsx_mid" in src
    n_ops = len(re.findall(r"// ---- P\d+ ", src))
    assert n_ops == len([l for l in api.model_describe(model_path(key)).splitlines() if re.match(r"P\d+ ", l)])      # one body per micro-op of the plan
    p = tmp_path / "mid.hip"
    p.write_text(src)
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-S", "-o", str(p) + ".s", str(p)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    asm = (tmp_path / "mid.hip.s").read_text()
    get = lambda k: int(re.search(r"; %s: *(\d+)" % k, asm).group(1))
    assert get("ScratchSize") <= SCRATCH_LIMIT[key], "register spills in the specialised kernel of %s: %d bytes" % (key, get("ScratchSize"))
    assert get("NumVgprs") <= 128
    assert int(re.search(r"LDSByteSize: *(\d+)", asm).group(1)) <= 160 * 1024
    assert "v_mfma_f32_16x16x4" in asm and "global_load_lds_dwordx4" in asm      # matrix cores and the LDS-DMA weight staging are really in there


@pytest.mark.parametrize("key", ["lite", "mlkit"])
def test_act16_variant_stores_arena_activations_as_halves(api, key, tmp_path, monkeypatch):
    """BSX_ACT16: same program, but every activation operand that lives in the arena is address space 3 (packed halves); pooled partial sums,
    gate vectors and weights stay f32.  It compiles within the same register budget and really contains 16-bit global accesses."""
    plain = api.model_kernel_source(model_path(key))
    monkeypatch.setenv("BSX_ACT16", "1")
    src = api.model_kernel_source(model_path(key))
    monkeypatch.delenv("BSX_ACT16")
    assert src != plain and re.search(r"_SP = 3\b", src) and not re.search(r"_SP = 3\b", plain)
    assert len(re.findall(r"_SP = 2\b", plain)) == len(re.findall(r"_SP = 3\b", src)) + len(re.findall(r"_SP = 2\b", src))
    p = tmp_path / "mid16.hip"
    p.write_text(src)
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-S", "-o", str(p) + ".s", str(p)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    asm = (tmp_path / "mid16.hip.s").read_text()
    get = lambda k: int(re.search(r"; %s: *(\d+)" % k, asm).group(1))
    assert get("ScratchSize") <= SCRATCH_LIMIT[key] and get("NumVgprs") <= 128
    assert "v_cvt_f16_f32" in asm or "v_cvt_pk" in asm


def test_arena_tensors_of_the_larger_graphs_are_staged_and_fused(api, monkeypatch, debug_switches):
    """MLKit / segm_full keep their 16x16x{96,128} (9x16) level-4 tensors in the arena.  The generator then (a) stages the depthwise inputs through the LDS
    workspace the planner reserved, chunk by chunk, and (b) where the 1x1 in front is the only producer computes each chunk straight into that workspace, so the
    expanded tensor is never stored; (c) 1x1 convolutions that write to the arena walk N-tile fastest.  segm_lite keeps those tensors in LDS: nothing to stage."""
    for key, fused in (("mlkit", 4), ("full", 4), ("lite", 0)):      # full (round 4): P2, P5, P9, P13 — the placement search fuses both level-3 pairs (the 88-channel one in 16-channel chunks with a ragged last one) and keeps the third level-4 block's 9x16x96 expanded tensor in LDS
        src = api.model_kernel_source(model_path(key))
        assert len(re.findall(r"computed chunk by chunk inside P\d+", src)) == fused, key
        assert ("op_pw<Op8_0>" in src) == (fused > 0)
        if fused:
            assert re.search(r"struct Op8_1 \{[^}]*N0 = \d+, NCOLS = (16|32), YSUB = \d+", src) and "NFAST = true" in src
        describe = api.model_describe(model_path(key))
        assert "lds_check=ok" in describe
    monkeypatch.setenv("BSX_RTC_NO_PWDW", "1")
    src = api.model_kernel_source(model_path("mlkit"))
    monkeypatch.delenv("BSX_RTC_NO_PWDW")
    assert "load_chunk<" in src and "computed chunk by chunk" not in src          # staging alone: the expanded tensor is loaded back from the arena
    monkeypatch.setenv("BSX_PLAN_NO_DW_STAGE", "1")
    src = api.model_kernel_source(model_path("mlkit"))
    assert "load_chunk<" not in src and "op_pw<Op8>" in src                        # the planner reserved no workspace: the plain ops


def test_precompile_fills_the_cache_without_a_gpu(api, tmp_path, monkeypatch):
    monkeypatch.setenv("BSX_KERNEL_CACHE", str(tmp_path))
    first = api.model_precompile(model_path("lite"))
    assert first.startswith("compiled") and str(tmp_path) in first
    assert [f for f in os.listdir(tmp_path) if f.endswith(".hsaco")]
    assert api.model_precompile(model_path("lite")).startswith("cached")
    # a graph without a per-frame program (DeepLab runs one launch per step) has nothing to specialise, and says so
    assert api.model_precompile(model_path("deeplab")).startswith("interpreted")
    assert api.model_kernel_source(model_path("deeplab")) == ""


def test_prelude_is_a_valid_translation_unit_on_its_own():
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-std=c++17", "--cuda-device-only", "-fsyntax-only", "-x", "hip",
                        os.path.join(here, "backscrub_amd", "csrc", "mid_prelude.hip")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]


def test_placement_search_keeps_the_cheapest_valid_plan(api, monkeypatch, debug_switches):
    """plan.cpp: build_frame_program lowers the program under eight placement policies and keeps the one that touches the fewest arena bytes per frame.  For every
    model and every forced policy the plan must pass the independent LDS check (no two live reservations overlap — also with the lifetimes an elided 1x1 extends),
    the default must be the minimum over the forced ones (ties: the lowest policy number), segm_lite / MLKit must stay on the round-3 plan (policy 0, i.e. the
    same kernels), and segm_full's search must really have paid: fewer arena tensors and < 60 % of policy 0's arena bytes."""
    def facts(key):
        line = [l for l in api.model_describe(model_path(key)).splitlines() if l.startswith("program ")][0]
        f = dict(kv.split("=") for kv in line.split()[1:])
        return int(f["arena_bytes_per_frame"]), int(f["placement_policy"]), int(f["hbm_tensors"]), f["lds_check"]
    for key in ("lite", "mlkit", "full"):
        forced = {}
        for pol in range(8):
            monkeypatch.setenv("BSX_PLAN_POLICY", str(pol))
            b, p, n_hbm, chk = facts(key)
            assert chk == "ok" and p == pol, (key, pol, chk)
            forced[pol] = (b, n_hbm)
        monkeypatch.delenv("BSX_PLAN_POLICY")
        b, p, n_hbm, chk = facts(key)
        best = min(v[0] for v in forced.values())
        assert chk == "ok" and b == best and p == min(q for q, v in forced.items() if v[0] == best), (key, b, p, forced)
        if key in ("lite", "mlkit"):
            assert p == 0
        else:
            assert p != 0 and n_hbm < forced[0][1] and b < 0.6 * forced[0][0], (b, forced[0])


def test_segment_lds_footprints_are_what_the_occupancy_story_says(api):
    """DESIGN 5.1 (round 4): seg_head without a stem tile / scratch block (<= 30 KB: 5 workgroups per CU), k3 / tail with the low-resolution window sized for the geometry
    (k3 <= 36 KB on segm_lite / segm_full: 4 per CU)."""
    import re
    for key, head_max, k3_max in (("lite", 30.0, 36.0), ("mlkit", 30.0, 46.0), ("full", 30.0, 36.0)):
        d = api.model_describe(model_path(key))
        kib = {m.group(1): float(m.group(2)) for m in re.finditer(r"segment (\w+)\s+tile \S+ \S+ tiles per frame, LDS ([\d.]+) KiB", d)}
        assert set(kib) == {"head", "k2", "k3", "tail"}, kib
        assert kib["head"] <= head_max and kib["k3"] <= k3_max and kib["tail"] <= 32.0, (key, kib)


def test_low_resolution_window_reservation_covers_every_tile(tmp_path):
    """segments.hpp: seg_lo_window_floats sizes the LDS window of the low-resolution tensor a k3 / tail tile interpolates from (round 4: the geometry's need, not a 15 KB worst
    case).  An under-reservation would let one tile's staged window run into the next LDS region, so the host function (compiled here from the header itself, -ffp-contract=off
    like the library) is held against a brute-force walk in float32 — every tile, every halo-region row and column, TFLite's index arithmetic — for the three models' geometries,
    both resize conventions, and odd sizes / tile shapes the planner could pick for other models."""
    import subprocess

    import numpy as np
    src = tmp_path / "lo.cpp"
    src.write_text('#include <cstdio>\n#include <cstdlib>\n#include "segments.hpp"\nint main(int c, char** v) { int a[10]; for (int i = 0; i < 10; i++) a[i] = atoi(v[i + 1]);\n'
                   '  printf("%d\\n", bsx::seg_lo_window_floats(a[0], a[1], a[2], a[3], a[4] != 0, a[5] != 0, a[6], a[7], a[8], a[9])); return 0; }\n')
    exe = tmp_path / "lo"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-I", os.path.join(ROOT, "backscrub_amd", "csrc"), str(src), "-o", str(exe)])

    import re
    stride = int(re.search(r"constexpr int kSegLoStride = (\d+);", open(os.path.join(ROOT, "backscrub_amd", "csrc", "segments.hpp")).read()).group(1))   # floats per staged pixel (16 since round 5: dense + swz_l)
    assert stride == 16

    def axis(o, scale, half_pixel, in_size):
        f = np.float32
        v = (f(o) + f(0.5)) * f(scale) + f(-0.5) if half_pixel else f(o) * f(scale)
        return max(int(np.floor(v)), 0), min(int(np.ceil(v)), in_size - 1)

    def brute(H, W, HL, WL, hp, al, TR, TC):
        f = np.float32
        hs = f(HL - 1) / f(H - 1) if (al and H > 1) else f(HL) / f(H)
        ws = f(WL - 1) / f(W - 1) if (al and W > 1) else f(WL) / f(W)
        ty_n, tx_n = -(-H // TR), -(-W // TC)
        need = 0
        for ty in range(ty_n):
            rows = range(max(ty * TR - 1, 0), min(ty * TR + TR, H - 1) + 1)
            lo_r = min(axis(r, hs, hp, HL)[0] for r in rows); hi_r = max(axis(r, hs, hp, HL)[1] for r in rows)
            for tx in range(tx_n):
                cols = range(max(tx * TC - 1, 0), min(tx * TC + TC, W - 1) + 1)
                lo_c = min(axis(c_, ws, hp, WL)[0] for c_ in cols); hi_c = max(axis(c_, ws, hp, WL)[1] for c_ in cols)
                need = max(need, (hi_r - lo_r + 1) * (hi_c - lo_c + 1) * stride)
        return need, ty_n, tx_n

    cases = [(48, 80, 24, 40, 16, 14), (24, 40, 12, 20, 12, 14),                 # segm_lite: tail, k3
             (128, 128, 64, 64, 16, 13), (64, 64, 32, 32, 16, 13),               # MLKit
             (72, 128, 36, 64, 18, 13), (36, 64, 18, 32, 12, 13),                # segm_full
             (50, 70, 25, 35, 17, 11), (33, 47, 17, 24, 9, 14), (40, 40, 13, 13, 18, 14)]   # odd sizes, a 3x up-sampling
    for H, W, HL, WL, TR, TC in cases:
        for hp, al in ((1, 0), (0, 0), (0, 1)):
            need, ty_n, tx_n = brute(H, W, HL, WL, bool(hp), bool(al), TR, TC)
            got = int(subprocess.check_output([str(exe)] + [str(x) for x in (H, W, HL, WL, hp, al, TR, TC, ty_n, tx_n)]).decode())
            assert got >= min(need, 12 * 16 * stride), ((H, W, HL, WL, hp, al, TR, TC), got, need)
            assert got <= 12 * 16 * stride


@pytest.mark.parametrize("key", ["lite", "full", "mlkit"])
def test_zero_cell_of_the_middle_kernel(api, key, monkeypatch, debug_switches):
    """Round 5: the last 16 bytes of the middle kernel's LDS block are a zero cell — the planner's blocks end below it, the kernel's prologue zeroes it, and the
    depthwise ops take their out-of-image taps from it (traits ZC = true) — since round 6 also the chunk-by-chunk ops, which round 5 had to leave out
    (profiles/r05j: the form cost MLKit's 128-register kernel 150 more bytes of spill; that kernel now needs 92 registers); BSX_RTC_NO_ZERO_CELL=1 (debug build)
    switches the form off everywhere."""
    hdr = open(os.path.join(ROOT, "backscrub_amd", "csrc", "frame_program.hpp")).read()
    total = eval(re.search(r"constexpr int kLdsTotalFloats = ([^;]+);", hdr).group(1))            # noqa: S307 — "160 * 256"
    assert total == 160 * 256 and "kLdsZeroOff = kLdsTotalFloats - kLdsZeroFloats" in hdr and "constexpr int kLdsZeroFloats = 4;" in hdr
    prelude = open(os.path.join(ROOT, "backscrub_amd", "csrc", "mid_prelude.hip")).read()
    # the cell's offset reaches the device templates from the PLAN (gen_mid.cpp emits BSXM_ZERO_OFF = Plan::lds_zero_off() in front of the prelude: ADVICE r5), the
    # literal in the prelude is only the stand-alone syntax check's default
    assert "constexpr int kZeroOff = BSXM_ZERO_OFF;" in prelude and "#define BSXM_ZERO_OFF (160 * 256 - 4)" in prelude and "kZeroOff - T::X_OFF" in prelude
    desc = api.model_describe(model_path(key))
    lds_floats = int(re.search(r"lds_floats=(\d+)", desc).group(1))
    assert lds_floats <= total - 4 and "lds_check=ok" in desc                                        # no planned block reaches into the cell
    src = api.model_kernel_source(model_path(key))
    assert "#define BSXM_LANES 1024\n#define BSXM_ZERO_OFF %d\n" % (total - 4) in src
    assert "static_assert(kZeroOff == %d && kThreads == 1024" % (total - 4) in src and "float smem[%d];" % total in src
    assert re.search(r"if \(threadIdx\.x < 4\) L\[kZeroOff \+ threadIdx\.x\] = 0\.f;", src)
    structs = re.findall(r"struct (Op\d+(?:_\d+)?) \{\n  static constexpr int K = [^\n]+\n  static constexpr bool ZC = (true|false);\n  static constexpr int X_SP = (\d), X_OFF", src)
    assert structs, "no depthwise traits found"
    for name, zc, sp in structs:
        assert zc == "true", (name, zc)                                                               # round 6: also the chunk-by-chunk ops (Op<i>_<c>) — their input is an LDS workspace
    assert any("_" in name for name, _, _ in structs) == (key != "lite")                              # (segm_lite keeps every depthwise input resident: no chunked op)
    monkeypatch.setenv("BSX_RTC_NO_ZERO_CELL", "1")
    off = api.model_kernel_source(model_path(key))
    assert "ZC = true" not in off and off.count("ZC = false") == len(structs)


def test_the_form_of_the_middle_kernel_is_chosen_by_its_scratch_size(api, monkeypatch, debug_switches):
    """build_mid_kernel (bsx_api.hip): the plain form first; where its code object reports scratch (kernel descriptor bytes 4-7, read from the ELF by
    rtc.cpp: code_object_scratch_bytes — no GPU), the form with the lane index behind an opaque asm is compiled too and taken if it spills less.  segm_lite and
    segm_full do not spill and keep the plain form (recomputing shared values costs them issue slots: profiles/r06g); MLKit's plain form spills 352 bytes, its
    opaque form none.  The forced forms of the debug build show the parser reads real values."""
    for key, opaque in (("lite", False), ("full", False), ("mlkit", True)):
        msg = api.model_precompile(model_path(key))
        assert " 0 B of scratch" in msg and ("lane indices re-derived per op" in msg) == opaque, (key, msg)
        assert ("#define BSXM_OPAQUE_TID 1" in api.model_kernel_source(model_path(key))) == opaque
    monkeypatch.setenv("BSX_RTC_TID", "0")
    plain = api.model_precompile(model_path("mlkit"))
    m = re.search(r"(\d+) B of scratch", plain)
    assert m and int(m.group(1)) > 0 and "re-derived" not in plain, plain
    monkeypatch.setenv("BSX_RTC_TID", "1")
    assert " 0 B of scratch, lane indices re-derived per op" in api.model_precompile(model_path("lite"))


@pytest.mark.parametrize("key", ["lite", "full", "mlkit"])
def test_segment_kernels_specialised_to_the_graph_compile_within_their_budgets(api, key, tmp_path):
    """Round 6 (csrc/gen_seg.cpp): the segment kernels are compiled a second time, by hipRTC when a context is created, with the loaded graph's descriptors and template
    arguments as compile-time constants.  Without a GPU: the source carries the plan's geometry, compiles for gfx950 (hipcc here, hipRTC through bsx_model_precompile),
    keeps the register steps of the ahead-of-time kernels (tests/test_kernel_resources.py: occupancy is what these kernels live on) and spills nothing."""
    src = api.model_seg_source(model_path(key))
    assert "#define BSXS_SEG_RTC 1" in src and "constexpr SegHead kSegHEAD" in src and "constexpr SegTail kSegTAIL" in src
    for k in ("bsx_seg_head", "bsx_seg_k2", "bsx_seg_k3", "bsx_seg_tail"):
        assert 'extern "C" __global__' in src and k in src
    desc = api.model_describe(model_path(key))
    m = re.search(r"segment head  tile (\d+)x(\d+), (\d+)x(\d+) tiles", desc)
    assert m and ("t.TR = %s; t.TC = %s; t.tiles_y = %s; t.tiles_x = %s;" % m.groups()) in src          # the constants ARE this plan's
    p = tmp_path / "seg.hip"
    p.write_text(src)
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-S", "-o", str(p) + ".s", str(p)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    asm = (tmp_path / "seg.hip.s").read_text()
    caps = {"bsx_seg_head": 88, "bsx_seg_k2": 80, "bsx_seg_k3": 104, "bsx_seg_tail": 96}
    for k, cap in caps.items():
        blk = asm[asm.index(".amdhsa_kernel " + k):]
        blk = blk[:blk.index(".end_amdhsa_kernel")]
        vg = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", blk).group(1))
        sc = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", blk).group(1))
        assert sc == 0 and vg <= cap, "%s of %s: %d registers (cap %d), %d B of scratch" % (k, key, vg, cap, sc)
    msg = api.model_precompile(model_path(key))
    assert re.search(r"segment kernels (compiled|cached) \(\d+ bytes of source, \d+ bytes of code object, 0 B of scratch\)", msg), msg
    assert api.model_seg_source(model_path("deeplab")) == ""                                          # per-launch path: no segment kernels
