"""The Rust shim crate (rust/all-is-cubes-hip) cannot be compiled here (no Rust toolchain); this keeps its FFI
declarations in step with include/aic_hip.h mechanically: same functions with the same number of parameters, same
struct fields in the same order, same constants."""
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
HEADER = (ROOT / "include" / "aic_hip.h").read_text()
FFI = (ROOT / "rust" / "all-is-cubes-hip" / "src" / "ffi.rs").read_text()


def strip_comments(text: str) -> str:
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def c_functions():
    out = {}
    for m in re.finditer(r"\b(aic_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", strip_comments(HEADER), flags=re.S):
        params = m.group(2).strip()
        out[m.group(1)] = 0 if params in ("", "void") else len(params.split(","))
    return out


def rust_functions():
    out = {}
    for m in re.finditer(r"pub fn (aic_[a-z0-9_]+)\s*\(([^)]*)\)", strip_comments(FFI), flags=re.S):
        params = [p for p in m.group(2).split(",") if p.strip()]
        out[m.group(1)] = len(params)
    return out


def test_every_entry_point_is_declared_with_the_same_arity():
    c, r = c_functions(), rust_functions()
    assert set(c) == set(r), (sorted(set(c) - set(r)), sorted(set(r) - set(c)))
    for name in c:
        assert c[name] == r[name], (name, c[name], r[name])


def c_structs():
    out = {}
    for m in re.finditer(r"typedef struct (aic_[a-z_]+)\s*\{(.*?)\}\s*\1\s*;", strip_comments(HEADER), flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                name = re.sub(r"\[[^\]]*\]", "", part.strip().split()[-1]).lstrip("*")
                fields.append(name)
        out[m.group(1)] = fields
    return out


def rust_structs():
    out = {}
    for m in re.finditer(r"pub struct (aic_[a-z_]+)\s*\{(.*?)\n\}", strip_comments(FFI), flags=re.S):
        out[m.group(1)] = re.findall(r"pub ([a-z_0-9]+)\s*:", m.group(2))
    return out


def test_struct_fields_match_in_order():
    c, r = c_structs(), rust_structs()
    for name, fields in c.items():
        assert name in r, name
        assert r[name] == fields, (name, fields, r[name])


def test_constants_match():
    for m in re.finditer(r"#define (AIC_[A-Z_]+) (\d+)u?\b", HEADER):
        name, value = m.group(1), int(m.group(2))
        rm = re.search(rf"pub const {name}: [a-z_0-9]+ = (\d+);", FFI)
        assert rm, name
        assert int(rm.group(1)) == value, name


def test_crate_files_exist():
    crate = ROOT / "rust" / "all-is-cubes-hip"
    for rel in ("Cargo.toml", "build.rs", "src/lib.rs", "src/ffi.rs", "src/flatten.rs"):
        assert (crate / rel).is_file(), rel
    lib = (crate / "src" / "lib.rs").read_text()
    assert "impl HeadlessRenderer for HipRtRenderer" in lib and "aic_upload_space" in lib and "aic_render" in lib
