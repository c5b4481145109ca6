"""Registers / scratch / code size of every op of a generated (graph-specialised) program, each compiled alone (no GPU needed).
   python tools/mid_op_regs.py [lite|full|mlkit]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from backscrub_amd import api  # noqa: E402
from conftest import model_path  # noqa: E402

key = sys.argv[1] if len(sys.argv) > 1 else "lite"
src = api.model_kernel_source(model_path(key))
head, kern = src.split('extern "C" __global__', 1)
sig, body = kern.split("{\n", 1)
pre = body.split("  // ---- P0", 1)[0]
ops = [o for o in re.split(r"(?=  // ---- P\d+ )", body[len(pre):]) if o.strip()]
# the FC weight loads of a squeeze-excite op are emitted one op early (they stay in registers across the op in front of it): for the
# isolated view they move back to their own op
for i in range(len(ops)):
    mine = re.findall(r"  FcRegs<[^\n]*> fc(\d+)_\d;\n  fc_load<[^\n]*\n", ops[i])
    for m in re.finditer(r"(  FcRegs<[^\n]*> fc(\d+)_\d;\n  fc_load<[^\n]*\n)", ops[i]):
        j = int(m.group(2))
        if j != i and j < len(ops):
            ops[i] = ops[i].replace(m.group(1), "")
            head_, rest_ = ops[j].split("\n", 1)
            ops[j] = head_ + "\n" + m.group(1) + rest_
tmp = tempfile.mkdtemp()
rows = []
for i, op in enumerate(ops):
    op = op.rsplit("  op_barrier();\n  if (tl && blockIdx.x == 0 && threadIdx.x == 0) tl[%d]" % (i + 1), 1)[0] if i == len(ops) - 1 else op
    op = re.sub(r"  stage<.*\n", "", op)                      # the next op's weight DMA belongs to the main loop, not to this body
    # a volatile LDS store keeps the compiler from treating the (here never written) LDS inputs of an isolated op as undefined
    one = head + 'extern "C" __global__' + sig + "{\n" + pre.split("  stage<")[0] + "  ((volatile lds_f*)L)[threadIdx.x] = 0.f;\n" + op + "}\n"
    p = os.path.join(tmp, "op%d.hip" % i)
    open(p, "w").write(one)
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-S", "-o", p + ".s", p],
                       capture_output=True, text=True)
    if r.returncode:
        print(r.stderr[-2000:])
        sys.exit(1)
    s = open(p + ".s").read()
    g = lambda k: int(re.search(r"; %s: *(\d+)" % k, s).group(1))
    code = int(re.search(r"codeLenInByte = (\d+)", s).group(1))
    rows.append((op.splitlines()[0].strip()[8:70], g("NumVgprs"), g("ScratchSize"), code, s.count("v_mfma"), s.count("ds_read") + s.count("ds_load"), s.count("s_waitcnt")))
print("%-64s %5s %7s %6s %5s %5s %5s" % ("op", "vgpr", "scratch", "bytes", "mfma", "dsrd", "waits"))
for r in rows:
    print("%-64s %5d %7d %6d %5d %5d %5d" % r)
print("asm in", tmp)
