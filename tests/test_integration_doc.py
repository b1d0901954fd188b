"""INTEGRATION.md shows the Rust `extern "C"` declarations a qdrant maintainer would paste; nothing compiles them here
(no rustc in the image), so this test parses them and checks name, arity and every parameter type against
include/qdrant_amd.h.  A wrong arity in that text is undefined behaviour for whoever binds it.  CPU only."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

C_SCALARS = {"int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "int64_t": "i64", "size_t": "usize", "float": "f32",
             "uint8_t": "u8", "char": "c_char", "void": "c_void"}


def _camel(name):
    if name == "qmx_scored_point":
        return "ScoredPointOffset"          # crosses the FFI verbatim (types.rs:12-17)
    return "".join(p.capitalize() for p in name.split("_"))


def _c_type(t):
    """C parameter type -> the Rust spelling the shim must use."""
    t = t.replace("volatile", " ").strip()
    stars = t.count("*")
    base = t.replace("*", " ")
    toks = base.split()
    # `const T *const *`: consts in order of appearance apply to successive pointer levels from the inside out
    consts = [tok == "const" for tok in toks]
    names = [tok for tok in toks if tok != "const"]
    assert len(names) == 1, t
    base_rs = C_SCALARS.get(names[0]) or _camel(names[0])
    if stars == 0:
        return base_rs
    # innermost pointee constness = a `const` before/after the base name; outer levels: `*const`
    inner_const = toks[0] == "const" or (len(toks) > 1 and toks[1] == "const" and toks[0] == names[0])
    n_outer_const = sum(consts) - (1 if inner_const else 0)
    out = base_rs
    quals = ["const" if inner_const else "mut"] + ["const" if i < n_outer_const else "mut" for i in range(stars - 1)]
    for q in quals:
        out = "*%s %s" % (q, out)
    return out


def _header_decls():
    text = open(os.path.join(ROOT, "include", "qdrant_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    out = {}
    for m in re.finditer(r"QMX_API\s+([\w\s\*]+?)\b(qmx_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"(.*?)(\w+)$", a)          # strip the parameter name
                params.append(_c_type(mm.group(1).strip()))
        out[name] = (None if ret == "void" else _c_type(ret), params)
    return out


def _rust_decls():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    decls = []
    for block in re.findall(r"```rust(.*?)```", text, flags=re.S):
        for ext in re.findall(r'unsafe extern "C" \{(.*?)\n\}', block, flags=re.S):
            ext = re.sub(r"//[^\n]*", "", ext)
            for m in re.finditer(r"fn\s+(qmx_\w+)\s*\(([^;]*?)\)\s*(?:->\s*([\w\s\*]+?))?\s*;", ext, flags=re.S):
                name, args, ret = m.group(1), " ".join(m.group(2).split()), m.group(3)
                params = [" ".join(a.split(":", 1)[1].split()) for a in args.split(",") if a.strip()]
                decls.append((name, ret.strip() if ret else None, params))
    return decls


def test_c_type_mapping():
    assert _c_type("const float *") == "*const f32"
    assert _c_type("qmx_segment **") == "*mut *mut QmxSegment"
    assert _c_type("const void *const *") == "*const *const c_void"
    assert _c_type("const volatile uint8_t *") == "*const u8"
    assert _c_type("qmx_scored_point *") == "*mut ScoredPointOffset"
    assert _c_type("const qmx_hnsw_build_params *") == "*const QmxHnswBuildParams"


def test_rust_declarations_match_the_header():
    header = _header_decls()
    rust = _rust_decls()
    assert len(rust) >= 40, "INTEGRATION.md lost its extern blocks"
    for name, ret, params in rust:
        assert name in header, "%s is declared in INTEGRATION.md but not in the header" % name
        hret, hparams = header[name]
        assert ret == hret, "%s: return type %s, header says %s" % (name, ret, hret)
        assert len(params) == len(hparams), "%s: %d parameters in INTEGRATION.md, %d in the header" % (name, len(params), len(hparams))
        for i, (a, b) in enumerate(zip(params, hparams)):
            assert a == b, "%s parameter %d: INTEGRATION.md has `%s`, the header `%s`" % (name, i, a, b)


def test_every_entry_point_is_shown_to_the_maintainer():
    declared = {n for n, _, _ in _rust_decls()}
    missing = sorted(set(_header_decls()) - declared)
    assert not missing, "entry points without a Rust declaration in INTEGRATION.md: %s" % missing
