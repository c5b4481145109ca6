"""The synthetic models must be architecture-identical to the reference's shipped files
(same operator multiset with the same shapes/options), and the python flatbuffer writer
must round-trip through all three readers (python, oracle C++, product C++)."""
import collections
import os

import numpy as np
import pytest

from conftest import MODEL_KEYS, ROOT
from backscrub_amd import tflite_io as T
from tools import make_synthetic_model as S


def _signature(m):
    sig = collections.Counter()
    for op in m.ops:
        ins = tuple(tuple(m.tensors[i].shape) for i in op.inputs if i >= 0)
        outs = tuple(tuple(m.tensors[o].shape) for o in op.outputs)
        opts = tuple(sorted((k, v) for k, v in op.opts.items() if k not in ("axis", "weights_format")))
        sig[(op.name, ins, outs, opts, op.custom)] += 1
    return sig


@pytest.mark.parametrize("key", list(MODEL_KEYS))
def test_synthetic_matches_reference_architecture(key):
    real = os.path.join(ROOT, "models", MODEL_KEYS[key])
    if not os.path.exists(real):
        pytest.skip("reference model not staged (only available where /root/reference exists)")
    a, b = _signature(T.load(real)), _signature(T.loads(T.dumps(S.build(key))))
    assert a == b, "missing %s\nextra %s" % (list((a - b).items())[:5], list((b - a).items())[:5])


@pytest.mark.parametrize("key", list(MODEL_KEYS))
def test_synthetic_runs_in_oracle_and_is_not_degenerate(oracle, key):
    path = S.ensure(key)
    m = T.load(path)
    om = oracle.Model(path)
    assert om.n_ops == len(m.ops)
    shp = om.shape(om.input)
    x = np.random.default_rng(0).uniform(0, 1, shp).astype(np.float32)
    y = om.invoke(x)
    assert np.isfinite(y).all()
    assert y.std() > 1e-3, "degenerate output"
    om.close()
