"""Regression digests of the CPU oracle (tests/golden/oracle_digests.json, written by tools/make_golden.py).
Self-generated — they freeze the oracle, they do not pin it to the reference (which ships no vectors)."""
import json
import os

from conftest import ROOT


def test_oracle_digests_unchanged(oracle):
    from tools import make_golden
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_digests.json")))
    got = make_golden.compute()
    assert got == want, "the oracle (or synth.py / the synthetic model generator) changed: re-run tools/make_golden.py if intended"
