"""Static checks of the rules the product path must keep: the CPU oracle is test infrastructure
(only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch it), nothing reads
the reference tree at run time, and there is no CPU fallback to route through."""
import ast
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _imports(path: Path):
    tree = ast.parse(path.read_text())
    for node in ast.walk(tree):
        if isinstance(node, ast.Import):
            for a in node.names:
                yield a.name, node.lineno
        elif isinstance(node, ast.ImportFrom):
            yield (node.module or ""), node.lineno


def _enclosing_function(path: Path, lineno: int):
    tree = ast.parse(path.read_text())
    best = None
    for node in ast.walk(tree):
        if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef)) and node.lineno <= lineno <= max(getattr(node, "end_lineno", node.lineno), node.lineno):
            if best is None or node.lineno > best.lineno:
                best = node
    return best.name if best else None


def test_package_never_imports_the_oracle():
    for path in (ROOT / "all_is_cubes_amd").rglob("*.py"):
        for mod, line in _imports(path):
            assert mod.split(".")[0] != "oracle", f"{path}:{line} imports the oracle"
    for path in list((ROOT / "all_is_cubes_amd").rglob("*.cpp")) + list((ROOT / "all_is_cubes_amd").rglob("*.hip")) + list(
            (ROOT / "all_is_cubes_amd").rglob("*.h*")) + list((ROOT / "include").glob("*.h")):
        text = path.read_text()
        assert "aic_oracle" not in text and "oracle/" not in text, f"{path} refers to the oracle"


def test_oracle_only_in_the_allowed_legs_of_the_root_scripts():
    for name, allowed in (("bench.py", {"cpu_baseline"}), ("__graft_entry__.py", {"smoke", "build"})):
        path = ROOT / name
        for mod, line in _imports(path):
            if mod.split(".")[0] == "oracle":
                assert _enclosing_function(path, line) in allowed, f"{name}:{line} imports the oracle outside {allowed}"
    # build() may compile the checker, but must not call into it
    src = (ROOT / "__graft_entry__.py").read_text()
    build_src = src[src.index("def build"):src.index("def smoke")]
    assert "import oracle" not in build_src


def test_nothing_reads_the_reference_tree_at_run_time():
    pat = re.compile(r"/root/reference")
    for path in [ROOT / "bench.py", ROOT / "__graft_entry__.py", *(ROOT / "all_is_cubes_amd").rglob("*.py"),
                 *(p for p in (ROOT / "tests").glob("*.py") if p.name != "test_product_boundaries.py")]:
        assert not pat.search(path.read_text()), f"{path} mentions /root/reference (only tests/golden/make_golden.py may)"


def test_no_cpu_render_path_in_the_package():
    # the package has no renderer of its own on the CPU: every draw goes through libaic_hip.so, whose
    # context creation fails without a device (tests/test_host_mirror.py checks the failure itself)
    abi_src = (ROOT / "all_is_cubes_amd" / "abi.py").read_text()
    assert "aic_create" in abi_src and "AicError" in abi_src
    host_src = (ROOT / "all_is_cubes_amd" / "host" / "aic_host.cpp").read_text()
    assert "aic_render" in host_src and "trace_ray" not in host_src
