"""Dependency-free reader and writer for TFLite (schema v3) flatbuffers.

Host-side tooling only: the product path parses models in C++
(`csrc/tflite_model.cpp`); this module exists so that tests can cross-check
that parser, so that PyTorch cross-checks can see the graph, and so that
`tools/make_synthetic_model.py` can emit architecture-faithful models with
seeded random weights (the reference's `.tflite` files do not travel to the
GPU box).

Field ids follow the public TFLite schema (`schema.fbs`, v3) as recorded in
SURVEY.md Appendix A.  Reference call site that consumes such files:
`/root/reference/lib/libbackscrub.cc:190` (FlatBufferModel::BuildFromFile).
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

# ---- builtin operator codes we know -------------------------------------------------
OPNAMES = {
    0: "ADD", 1: "AVERAGE_POOL_2D", 2: "CONCATENATION", 3: "CONV_2D",
    4: "DEPTHWISE_CONV_2D", 6: "DEQUANTIZE", 9: "FULLY_CONNECTED", 14: "LOGISTIC",
    18: "MUL", 19: "RELU", 21: "RELU6", 22: "RESHAPE", 23: "RESIZE_BILINEAR",
    25: "SOFTMAX", 32: "CUSTOM", 117: "HARD_SWISH",
}
OPCODES = {v: k for k, v in OPNAMES.items()}
# builtin_options union tags (schema.fbs BuiltinOptions)
OPT_CONV2D, OPT_DWCONV, OPT_POOL, OPT_FC, OPT_CONCAT, OPT_ADD, OPT_RESIZE_BILINEAR, OPT_MUL = 1, 2, 5, 8, 10, 11, 15, 21
OPT_DEQUANTIZE, OPT_HARD_SWISH = 26, 91
TENSOR_F32, TENSOR_F16, TENSOR_I32 = 0, 1, 2
_NP = {TENSOR_F32: np.float32, TENSOR_F16: np.float16, TENSOR_I32: np.int32}


@dataclass
class Tensor:
    shape: List[int]
    type: int
    buffer: int
    name: str
    data: Optional[np.ndarray] = None  # constant payload (None for activations)


@dataclass
class Op:
    code: int                 # builtin code
    name: str                 # OPNAMES entry or custom_code
    inputs: List[int]
    outputs: List[int]
    opts: Dict[str, int] = field(default_factory=dict)
    custom: bytes = b""


@dataclass
class Model:
    tensors: List[Tensor]
    ops: List[Op]
    inputs: List[int]
    outputs: List[int]
    description: str = ""


# ======================================================================================
# reader
# ======================================================================================
class _FB:
    def __init__(self, buf: bytes):
        self.b = buf

    def u8(self, o): return self.b[o]
    def i8(self, o): return struct.unpack_from("<b", self.b, o)[0]
    def u16(self, o): return struct.unpack_from("<H", self.b, o)[0]
    def i32(self, o): return struct.unpack_from("<i", self.b, o)[0]
    def u32(self, o): return struct.unpack_from("<I", self.b, o)[0]

    def field(self, tbl, idx):
        """absolute offset of field `idx` inside table `tbl`, or 0 if defaulted"""
        vt = tbl - self.i32(tbl)
        vsz = self.u16(vt)
        slot = 4 + 2 * idx
        if slot >= vsz:
            return 0
        off = self.u16(vt + slot)
        return tbl + off if off else 0

    def indirect(self, o): return o + self.u32(o)

    def scalar(self, tbl, idx, fmt, default=0):
        o = self.field(tbl, idx)
        return struct.unpack_from(fmt, self.b, o)[0] if o else default

    def table(self, tbl, idx):
        o = self.field(tbl, idx)
        return self.indirect(o) if o else 0

    def vec(self, tbl, idx):
        """(start, length) of a vector field"""
        o = self.field(tbl, idx)
        if not o:
            return 0, 0
        v = self.indirect(o)
        return v + 4, self.u32(v)

    def string(self, tbl, idx):
        s, n = self.vec(tbl, idx)
        return self.b[s:s + n].decode("utf-8", "replace") if s else ""

    def vec_i32(self, tbl, idx):
        s, n = self.vec(tbl, idx)
        return list(struct.unpack_from("<%di" % n, self.b, s)) if n else []

    def vec_tables(self, tbl, idx):
        s, n = self.vec(tbl, idx)
        return [self.indirect(s + 4 * i) for i in range(n)]


def _read_opts(fb: _FB, op_tbl: int, code: int) -> Dict[str, int]:
    t = fb.table(op_tbl, 4)
    if not t:
        return {}
    g = lambda i, fmt="<b", d=0: fb.scalar(t, i, fmt, d)
    if code == OPCODES["CONV_2D"]:
        return dict(padding=g(0), stride_w=g(1, "<i"), stride_h=g(2, "<i"), act=g(3),
                    dil_w=g(4, "<i", 1), dil_h=g(5, "<i", 1))
    if code == OPCODES["DEPTHWISE_CONV_2D"]:
        return dict(padding=g(0), stride_w=g(1, "<i"), stride_h=g(2, "<i"), depth_mult=g(3, "<i"),
                    act=g(4), dil_w=g(5, "<i", 1), dil_h=g(6, "<i", 1))
    if code == OPCODES["AVERAGE_POOL_2D"]:
        return dict(padding=g(0), stride_w=g(1, "<i"), stride_h=g(2, "<i"), filter_w=g(3, "<i"),
                    filter_h=g(4, "<i"), act=g(5))
    if code == OPCODES["FULLY_CONNECTED"]:
        return dict(act=g(0), weights_format=g(1), keep_num_dims=g(2))
    if code == OPCODES["CONCATENATION"]:
        return dict(axis=g(0, "<i"), act=g(1))
    if code in (OPCODES["ADD"], OPCODES["MUL"]):
        return dict(act=g(0))
    if code == OPCODES["RESIZE_BILINEAR"]:
        return dict(align_corners=g(2), half_pixel_centers=g(3))
    return {}


def load(path: str) -> Model:
    with open(path, "rb") as f:
        buf = f.read()
    return loads(buf)


def loads(buf: bytes) -> Model:
    fb = _FB(buf)
    root = fb.indirect(0)
    opcodes = []
    for oc in fb.vec_tables(root, 1):
        dep = fb.scalar(oc, 0, "<b", 0)
        full = fb.scalar(oc, 3, "<i", 0)
        opcodes.append((max(dep, full), fb.string(oc, 1)))
    buffers = []
    for bt in fb.vec_tables(root, 4):
        s, n = fb.vec(bt, 0)
        buffers.append((s, n))
    sg = fb.vec_tables(root, 2)[0]
    tensors = []
    for tt in fb.vec_tables(sg, 0):
        shape = fb.vec_i32(tt, 0)
        ttype = fb.scalar(tt, 1, "<b", 0)
        bidx = fb.scalar(tt, 2, "<I", 0)
        t = Tensor(shape, ttype, bidx, fb.string(tt, 3))
        s, n = buffers[bidx] if bidx < len(buffers) else (0, 0)
        if n and ttype in _NP:
            t.data = np.frombuffer(buf, dtype=_NP[ttype], count=n // np.dtype(_NP[ttype]).itemsize,
                                   offset=s).reshape(shape if shape else [-1]).copy()
        tensors.append(t)
    ops = []
    for ot in fb.vec_tables(sg, 3):
        code, cname = opcodes[fb.scalar(ot, 0, "<I", 0)]
        s, n = fb.vec(ot, 5)
        ops.append(Op(code, cname if code == 32 else OPNAMES.get(code, "OP%d" % code),
                      fb.vec_i32(ot, 1), fb.vec_i32(ot, 2), _read_opts(fb, ot, code), bytes(buf[s:s + n])))
    return Model(tensors, ops, fb.vec_i32(sg, 1), fb.vec_i32(sg, 2), fb.string(root, 3))


# ======================================================================================
# writer — a minimal back-to-front flatbuffer builder
# ======================================================================================
class Builder:
    """Builds the buffer back-to-front like the canonical flatbuffers builder:
    `self.buf` holds the *tail* of the file; an object's "offset" is its
    distance from the end of the file, so it stays valid while we prepend."""

    def __init__(self):
        self.buf = bytearray()
        self.minalign = 1

    def _pad(self, n):
        self.buf[0:0] = bytes(n)

    def prep(self, size, additional=0):
        """make sure that after writing `additional` bytes, the next `size`-byte scalar is aligned"""
        self.minalign = max(self.minalign, size)
        pad = (-(len(self.buf) + additional)) % size
        self._pad(pad)

    def put(self, fmt, v):
        self.buf[0:0] = struct.pack(fmt, v)

    def off(self):
        return len(self.buf)

    def put_uoffset(self, target):
        self.prep(4)
        self.put("<I", self.off() + 4 - target)

    def bytes_vec(self, data: bytes, align=16):
        self.prep(4, len(data))
        self.prep(align, len(data))  # data start aligned (length prefix sits just before)
        self.buf[0:0] = data
        self.put("<I", len(data))
        return self.off()

    def string(self, s: str):
        d = s.encode() + b"\0"
        self.prep(4, len(d))
        self.buf[0:0] = d
        self.put("<I", len(d) - 1)
        return self.off()

    def vec_i32(self, vals):
        self.prep(4, 4 * len(vals))
        for v in reversed(vals):
            self.put("<i", v)
        self.put("<I", len(vals))
        return self.off()

    def vec_offsets(self, offs):
        self.prep(4, 4 * len(offs))
        for o in reversed(offs):
            self.put_uoffset(o)
        self.put("<I", len(offs))
        return self.off()

    def table(self, fields):
        """fields: list of (idx, kind, value) with kind in {'i8','u8','i32','u32','off'};
        offsets must have been created before calling."""
        sizes = {"i8": 1, "u8": 1, "bool": 1, "i32": 4, "u32": 4, "off": 4}
        fmts = {"i8": "<b", "u8": "<B", "bool": "<B", "i32": "<i", "u32": "<I"}
        nf = (max(f[0] for f in fields) + 1) if fields else 0
        slots = [0] * nf
        start_end = self.off()
        # write fields, largest first for tidy alignment
        pos = {}
        for idx, kind, val in sorted(fields, key=lambda f: -sizes[f[1]]):
            if kind == "off":
                self.put_uoffset(val)
            else:
                self.prep(sizes[kind])
                self.put(fmts[kind], val)
            pos[idx] = self.off()
        self.prep(4)
        self.put("<i", 0)  # soffset placeholder
        tbl = self.off()
        tbl_size = tbl - start_end
        for idx, p in pos.items():
            slots[idx] = tbl - p
        # vtable
        vt = struct.pack("<HH", 4 + 2 * nf, tbl_size) + b"".join(struct.pack("<H", s) for s in slots)
        if len(vt) % 4:
            # keep table 4-aligned: pad *before* the vtable (i.e. at lower address)
            pass
        self.buf[0:0] = vt
        vt_off = self.off()
        if len(self.buf) % 2:
            raise AssertionError("vtable misaligned")
        # patch soffset: table_pos - vtable_pos (positive, vtable is at lower address)
        tpos = len(self.buf) - tbl
        struct.pack_into("<i", self.buf, tpos, vt_off - tbl)
        # re-align to 4 so later objects see an aligned tail start
        pad = (-len(self.buf)) % 4
        # padding at the front would shift nothing that's already placed (offsets are from the end)
        self._pad(pad)
        return tbl

    def finish(self, root, ident=b"TFL3"):
        self.prep(self.minalign, 8)
        self.buf[0:0] = ident
        self.put("<I", self.off() + 4 - root)
        return bytes(self.buf)


def _opts_table(b: Builder, op: Op):
    o = op.opts
    c = op.code
    if c == OPCODES["CONV_2D"]:
        return OPT_CONV2D, b.table([(0, "i8", o.get("padding", 0)), (1, "i32", o.get("stride_w", 1)),
                                    (2, "i32", o.get("stride_h", 1)), (3, "i8", o.get("act", 0)),
                                    (4, "i32", o.get("dil_w", 1)), (5, "i32", o.get("dil_h", 1))])
    if c == OPCODES["DEPTHWISE_CONV_2D"]:
        return OPT_DWCONV, b.table([(0, "i8", o.get("padding", 0)), (1, "i32", o.get("stride_w", 1)),
                                    (2, "i32", o.get("stride_h", 1)), (3, "i32", o.get("depth_mult", 1)),
                                    (4, "i8", o.get("act", 0)), (5, "i32", o.get("dil_w", 1)),
                                    (6, "i32", o.get("dil_h", 1))])
    if c == OPCODES["AVERAGE_POOL_2D"]:
        return OPT_POOL, b.table([(0, "i8", o.get("padding", 1)), (1, "i32", o.get("stride_w", 1)),
                                  (2, "i32", o.get("stride_h", 1)), (3, "i32", o["filter_w"]),
                                  (4, "i32", o["filter_h"]), (5, "i8", o.get("act", 0))])
    if c == OPCODES["FULLY_CONNECTED"]:
        return OPT_FC, b.table([(0, "i8", o.get("act", 0)), (1, "i8", 0), (2, "bool", o.get("keep_num_dims", 0))])
    if c == OPCODES["CONCATENATION"]:
        return OPT_CONCAT, b.table([(0, "i32", o.get("axis", 3)), (1, "i8", o.get("act", 0))])
    if c == OPCODES["ADD"]:
        return OPT_ADD, b.table([(0, "i8", o.get("act", 0))])
    if c == OPCODES["MUL"]:
        return OPT_MUL, b.table([(0, "i8", o.get("act", 0))])
    if c == OPCODES["RESIZE_BILINEAR"]:
        return OPT_RESIZE_BILINEAR, b.table([(2, "bool", o.get("align_corners", 0)),
                                             (3, "bool", o.get("half_pixel_centers", 0))])
    return 0, 0


def dumps(m: Model) -> bytes:
    """Serialise a Model. Every constant tensor gets its own buffer; buffer 0 is the
    conventional empty buffer."""
    b = Builder()
    # ---- buffers (written first => they end up at the file tail, like real models)
    buf_offs = []
    blobs = [b""]
    tensor_buf = []
    for t in m.tensors:
        if t.data is not None:
            blobs.append(np.ascontiguousarray(t.data, dtype=_NP[t.type]).tobytes())
            tensor_buf.append(len(blobs) - 1)
        else:
            tensor_buf.append(0)
    for blob in blobs:
        if blob:
            d = b.bytes_vec(blob)
            buf_offs.append(b.table([(0, "off", d)]))
        else:
            buf_offs.append(b.table([]))
    buffers_vec = b.vec_offsets(buf_offs)
    # ---- operator codes
    codes = []
    code_index = {}
    for op in m.ops:
        key = (op.code, op.name if op.code == 32 else "")
        if key not in code_index:
            code_index[key] = len(codes)
            codes.append(key)
    oc_offs = []
    for code, cname in codes:
        f = [(0, "i8", min(code, 127)), (2, "i32", 1), (3, "i32", code)]
        if cname:
            f.append((1, "off", b.string(cname)))
        oc_offs.append(b.table(f))
    opcodes_vec = b.vec_offsets(oc_offs)
    # ---- tensors
    t_offs = []
    for t, bi in zip(m.tensors, tensor_buf):
        name = b.string(t.name)
        shape = b.vec_i32(t.shape)
        t_offs.append(b.table([(0, "off", shape), (1, "i8", t.type), (2, "u32", bi), (3, "off", name)]))
    tensors_vec = b.vec_offsets(t_offs)
    # ---- operators
    o_offs = []
    for op in m.ops:
        f = []
        if op.custom:
            f.append((5, "off", b.bytes_vec(op.custom, align=4)))
        ty, ot = _opts_table(b, op)
        if ot:
            f += [(3, "u8", ty), (4, "off", ot)]
        outs = b.vec_i32(op.outputs)
        ins = b.vec_i32(op.inputs)
        f += [(0, "u32", code_index[(op.code, op.name if op.code == 32 else "")]), (1, "off", ins), (2, "off", outs)]
        o_offs.append(b.table(f))
    ops_vec = b.vec_offsets(o_offs)
    sg_name = b.string("main")
    sg_out = b.vec_i32(m.outputs)
    sg_in = b.vec_i32(m.inputs)
    sg = b.table([(0, "off", tensors_vec), (1, "off", sg_in), (2, "off", sg_out), (3, "off", ops_vec), (4, "off", sg_name)])
    sgs = b.vec_offsets([sg])
    desc = b.string(m.description or "backscrub_amd synthetic model")
    root = b.table([(0, "u32", 3), (1, "off", opcodes_vec), (2, "off", sgs), (3, "off", desc), (4, "off", buffers_vec)])
    return b.finish(root)


def save(m: Model, path: str):
    with open(path, "wb") as f:
        f.write(dumps(m))


def summarize(m: Model) -> str:
    lines = []
    for i, op in enumerate(m.ops):
        ins = ",".join("%d%s" % (t, m.tensors[t].shape) for t in op.inputs if t >= 0)
        outs = ",".join("%d%s" % (t, m.tensors[t].shape) for t in op.outputs)
        lines.append("#%d %s in[%s] out[%s] %s" % (i, op.name, ins, outs, op.opts or ""))
    return "\n".join(lines)


if __name__ == "__main__":
    import sys
    mm = load(sys.argv[1])
    print("inputs", mm.inputs, "outputs", mm.outputs, "ntensors", len(mm.tensors), "nops", len(mm.ops))
    print(summarize(mm))
