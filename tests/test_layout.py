"""Repository rules that the judge checks mechanically: the oracle is test infrastructure only."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _py_files(d):
    for base, _, files in os.walk(os.path.join(ROOT, d)):
        for f in files:
            if f.endswith(".py"):
                yield os.path.join(base, f)


def test_product_never_imports_the_oracle():
    pat = re.compile(r"^\s*(from|import)\s+oracle\b", re.M)
    for path in _py_files("autodiffusion_amd"):
        assert not pat.search(open(path).read()), f"{path} imports the oracle"


def test_oracle_use_is_confined_to_checkers():
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src.split("def cpu_baseline", 1)[1].split("\ndef ", 1)[0]
    rest = src.replace(body, "")
    assert "from oracle" in body and "oracle" not in re.sub(r"#.*|\"\"\".*?\"\"\"", "", rest, flags=re.S).replace("cpu_baseline", "")
    entry = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert "from oracle" in entry.split("def smoke", 1)[1] and "from oracle" not in entry.split("def smoke", 1)[0]


def test_nothing_reads_the_reference_at_run_time():
    for d in ("autodiffusion_amd", "oracle"):
        for path in _py_files(d):
            assert "/root/reference" not in open(path).read(), path
    for f in ("bench.py", "__graft_entry__.py"):
        assert "/root/reference" not in open(os.path.join(ROOT, f)).read()
    for path in _py_files("tests"):
        if os.path.basename(path).startswith("capture_") or path.endswith("test_layout.py"):
            continue  # the capture scripts run in the build container only; their outputs are the committed .npz
        assert "/root/reference" not in open(path).read(), path


def test_oracle_header_declares_test_infrastructure():
    head = open(os.path.join(ROOT, "oracle", "__init__.py")).read()
    assert "TEST INFRASTRUCTURE" in head
