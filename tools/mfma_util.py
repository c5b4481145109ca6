"""MFMA-pipe utilisation per kernel from a tools/pmc_sq2.sh summary.
Usage: python tools/mfma_util.py gpurun_out/pmc_sq2_<tag>.md
util = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs); kernel cycles = SQ_BUSY_CYCLES / 32 (the counter is summed
over the chip's 32 shader engines' sequencers; checked against the kernel-trace duration x the shader clock)."""
import collections
import re
import subprocess
import sys

rows = collections.defaultdict(dict)
for line in open(sys.argv[1]):
    m = re.match(r"\| (\S.*?) \| (SQ_\w+) \| (\d+) \| ([\d.]+) \|", line)
    if m:
        rows[m.group(1)][m.group(2)] = (int(m.group(3)), float(m.group(4)))


def pretty(name):
    if not name.startswith("_Z"):
        return name
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name + "fS_i"], capture_output=True, text=True).stdout
        m = re.search(r"(\w+<[^>]*>)", out)
        if m:
            return m.group(1)
    except OSError:
        pass
    m = re.match(r"_ZN3bsx12_GLOBAL__N_1\d+(\w+?_k)I(.*?)EEv", name)
    if m:
        args = re.findall(r"L([ib])(\d+)E", m.group(2))
        return "%s<%s>" % (m.group(1), ",".join(a[1] for a in args))
    return name


print("| kernel | launches | kernel cycles (avg) | MFMA busy cycles (avg, all SIMDs) | MFMA utilisation | VALU insts / wave-quad-cycle | wait-any share |")
print("|---|---:|---:|---:|---:|---:|---:|")
out = []
for k, c in rows.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "SQ_BUSY_CYCLES" not in c or c["SQ_VALU_MFMA_BUSY_CYCLES"][1] == 0:
        continue
    cyc = c["SQ_BUSY_CYCLES"][1] / 32.0
    mf = c["SQ_VALU_MFMA_BUSY_CYCLES"][1]
    wc = c.get("SQ_WAVE_CYCLES", (0, 0))[1]
    out.append((cyc * c["SQ_BUSY_CYCLES"][0], "| %s | %d | %.0f | %.0f | %.1f%% | %.3f | %.0f%% |" % (
        pretty(k), c["SQ_BUSY_CYCLES"][0], cyc, mf, 100 * mf / (cyc * 1024),
        c.get("SQ_INSTS_VALU", (0, 0))[1] / wc if wc else 0, 100 * c.get("SQ_WAIT_ANY", (0, 0))[1] / wc if wc else 0)))
for _, l in sorted(out, reverse=True):
    print(l)
