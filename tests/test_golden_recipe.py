"""The fixture recipe is part of the pin: `oracle/make_golden.py` must (a) run at HEAD, (b) bind the
REFERENCE's classes (not the repository's top-level `models/` / `modules/` alias packages of the same
names), and (c) reproduce `tests/golden/*.npz` bit for bit.  Needs /root/reference, i.e. runs in the
build container only (the GPU box has no reference: skipped there)."""
import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
GOLDEN = os.path.join(REPO, "tests", "golden")

needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "modules")), reason="the reference tree is not on this box")


def _run(args, cwd):
    env = dict(os.environ, PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""))  # the worst case: the alias packages first on the path
    return subprocess.run([sys.executable, os.path.join(REPO, "oracle", "make_golden.py")] + args, cwd=cwd, env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)


def _same(a, b):
    return a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes()


@needs_ref
def test_recipe_regenerates_every_fixture_bit_for_bit(tmp_path):
    r = _run(["--out", str(tmp_path)], cwd=REPO)   # cwd = repo root: `models/`, `modules/` of the repo are importable
    assert r.returncode == 0, r.stdout[-3000:]
    made, kept = sorted(os.listdir(tmp_path)), sorted(f for f in os.listdir(GOLDEN) if f.endswith(".npz"))
    assert made == kept, set(made) ^ set(kept)
    arrays = 0
    for f in made:
        x, y = np.load(tmp_path / f), np.load(os.path.join(GOLDEN, f))
        assert sorted(x.files) == sorted(y.files), f
        for k in x.files:
            assert _same(x[k], y[k]), f"{f}:{k} is not reproduced by oracle/make_golden.py"
            arrays += 1
    assert arrays > 5000


@needs_ref
@pytest.mark.parametrize("group,prefix", [("g1_g2_convs", "G1_"), ("g8_model", "G8_"), ("g12_pna", "G12_")])
def test_recipe_regenerates_a_selected_group(tmp_path, group, prefix):
    r = _run([group, "--out", str(tmp_path)], cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-3000:]
    made = [f for f in sorted(os.listdir(tmp_path)) if f.startswith(prefix)]
    assert made, os.listdir(tmp_path)
    for f in made:
        x, y = np.load(tmp_path / f), np.load(os.path.join(GOLDEN, f))
        for k in y.files:
            assert _same(x[k], y[k]), f"{f}:{k}"


@needs_ref
def test_recipe_binds_the_reference_classes_not_the_alias_packages():
    code = ("import sys, os, runpy; sys.argv=['make_golden.py','--help-none'];"
            "import importlib.util as u; s=u.spec_from_file_location('mg', os.path.join(%r,'oracle','make_golden.py'));"
            "m=u.module_from_spec(s); s.loader.exec_module(m); import inspect;"
            "print(inspect.getsourcefile(m.GCNConv)); print(inspect.getsourcefile(m.GNNTransformer)); print(inspect.getsourcefile(m.pad_batch))") % REPO
    env = dict(os.environ, PYTHONPATH=REPO)
    r = subprocess.run([sys.executable, "-c", code], cwd=REPO, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    files = r.stdout.strip().splitlines()[-3:]
    assert all(f.startswith(REF + os.sep) for f in files), files
