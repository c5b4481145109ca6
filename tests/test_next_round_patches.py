"""tools/next_round/ holds patches that were prepared (compiled, assembly counted) after a round's GPU budget was spent and are NOT applied.  They are only worth
keeping while they still apply to the tree they were written against."""
import glob
import os
import subprocess

import pytest

from conftest import ROOT


@pytest.mark.parametrize("patch", sorted(glob.glob(os.path.join(ROOT, "tools", "next_round", "*.patch"))) or [None])
def test_prepared_patch_still_applies(patch):
    if patch is None:
        pytest.skip("no prepared patches")
    if not os.path.isdir(os.path.join(ROOT, ".git")):
        pytest.skip("not a git checkout")
    r = subprocess.run(["git", "apply", "--check", patch], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, "%s no longer applies: %s" % (os.path.basename(patch), r.stderr[-400:])
