"""CPU: the C-ABI library loads, exports every symbol ``include/hf_pcg.h`` declares,
and the product never imports the oracle or falls back to the CPU."""

import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "hf_pcg.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hf_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from pytorchhessianfree_amd import _lib

    lib = _lib.load()
    names = _declared_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), name
        assert name in _lib.SIGNATURES, f"{name} is declared but not bound in _lib.py"
    assert lib.hf_abi_version() == _lib.ABI_VERSION
    assert lib.hf_error_string(-2).decode().startswith("hf:")


def test_bad_arguments_are_rejected_without_a_gpu():
    from pytorchhessianfree_amd import _lib

    lib = _lib.load()
    assert lib.hf_pcg_create(None, 10, 0, 0) == -1
    assert lib.hf_pcg_iterate(None, None, 0.0, None) == -1
    assert lib.hf_pack(None, None, None, None, 0, 1.0, 0, 0, None) == -1
    assert lib.hf_allreduce_sum(None, None, 1, 0, None) == -1


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "pytorchhessianfree_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
    bench = open(os.path.join(ROOT, "bench.py")).read()
    # bench.py may import the oracle only inside cpu_baseline()
    body = bench.split("def cpu_baseline")[1].split("\ndef ")[0]
    assert bench.count("from oracle") == body.count("from oracle") > 0


def test_cpu_tensors_are_refused_loudly():
    import pytorchhessianfree_amd as hf

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        hf.cg(lambda v: v, torch.ones(8))
    model = torch.nn.Linear(3, 2)
    opt = hf.HessianFree(model.parameters())
    x, t = torch.rand(4, 3), torch.rand(4, 2)

    def forward():
        out = model(x)
        return torch.nn.functional.mse_loss(out, t), out

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        opt.step(forward)


def test_missing_library_is_an_error(monkeypatch):
    from pytorchhessianfree_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libhfpcg.so")
    with pytest.raises(RuntimeError, match="REQUIRED"):
        _lib.load()


def test_every_relative_import_of_the_package_resolves():
    """Function-level ``from .x import y`` lines are only executed on a GPU box: a module moved into a sub-package
    (``engine/``) must not leave a stale relative import behind for the GPU run to find."""
    import ast
    import glob
    import importlib
    import importlib.util

    root = os.path.join(ROOT, "pytorchhessianfree_amd")
    for path in glob.glob(os.path.join(root, "**", "*.py"), recursive=True):
        rel = os.path.relpath(path, ROOT)[:-3].split(os.sep)
        pkg = rel[:-1] if rel[-1] != "__init__" else rel[:-1]
        for node in ast.walk(ast.parse(open(path).read())):
            if isinstance(node, ast.ImportFrom) and node.level > 0:
                base = pkg[: len(pkg) - (node.level - 1)]
                mod = ".".join(base + ([node.module] if node.module else []))
                m = importlib.import_module(mod)
                for a in node.names:
                    assert hasattr(m, a.name) or importlib.util.find_spec(mod + "." + a.name) is not None, (path, mod, a.name)


def test_environment_switches_are_registered():
    """Every ``HF_*`` environment variable the package reads is listed ONCE in ``config.SWITCHES`` with its default,
    its meaning and the test that exercises a non-default value; that test exists and names the switch."""
    import glob

    from pytorchhessianfree_amd import config

    pat = re.compile(r'(?:environ\.get\(|environ\[|getenv\(|env\.get\(|env\.setdefault\()\s*"(HF_[A-Z0-9_]+)"')
    read = {}
    root = os.path.join(ROOT, "pytorchhessianfree_amd")
    for ext in ("py", "hip", "h"):
        for path in glob.glob(os.path.join(root, "**", "*." + ext), recursive=True):
            for name in pat.findall(open(path).read()):
                read.setdefault(name, set()).add(os.path.relpath(path, ROOT))
    assert set(read) <= set(config.SWITCHES), {k: v for k, v in read.items() if k not in config.SWITCHES}
    assert set(config.SWITCHES) <= set(read), sorted(set(config.SWITCHES) - set(read))  # (no stale entries)
    for name, (default, meaning, test) in config.SWITCHES.items():
        path, func = test.split("::")
        text = open(os.path.join(ROOT, path)).read()
        assert f"def {func}(" in text, (name, test)
        assert re.search(r"\b" + name + r"\b", text), f"{test} does not mention {name}"
        assert meaning and isinstance(default, str)


def test_alternate_library_path(tmp_path):
    """``HF_PCG_LIB``: another build of the library (tuning variants) is loaded from the given path -- and a path
    that does not exist is an error, not a fall-back to the in-tree build."""
    import shutil
    import subprocess
    import sys

    from pytorchhessianfree_amd import _lib

    copy = tmp_path / "libhfpcg_variant.so"
    shutil.copyfile(_lib.LIB_PATH, copy)
    code = ("import sys; sys.path.insert(0, %r); from pytorchhessianfree_amd import _lib; "
            "print(_lib.LIB_PATH); print(_lib.load().hf_abi_version())" % ROOT)
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, HF_PCG_LIB=str(copy)), capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert p.stdout.split()[-2:] == [str(copy), str(_lib.ABI_VERSION)]
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, HF_PCG_LIB=str(tmp_path / "missing.so")),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
