"""The Rust binding a maintainer of the reference links (rust/bn254-verify-amd-sys) against the C ABI it binds (include/bn254_verify.h): both files
are parsed here, independently of the generator (tools/gen_rust_sys.py), and every function's name, arity, return type and per-parameter type class -- pointer
depth and constness, pointee kind, integer width -- must agree, as must every constant.  No Rust toolchain exists in this image (rust/README.md): this comparison is
what stands in for `cargo build`."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C_KIND = {"int": "i32", "unsigned": "u32", "size_t": "usize", "uint64_t": "u64", "long": "long", "float": "f32", "double": "f64", "uint8_t": "u8", "void": "void", "char": "char",
          "bn254_g16_pvk": "opaque:g16", "bn254_plonk_pvk": "opaque:plonk"}
R_KIND = {"c_int": "i32", "c_uint": "u32", "usize": "usize", "u64": "u64", "c_long": "long", "f32": "f32", "f64": "f64", "u8": "u8", "c_void": "void", "c_char": "char",
          "Bn254G16Pvk": "opaque:g16", "Bn254PlonkPvk": "opaque:plonk", "()": "void"}


def _c_class(ty, array):
    toks = ty.replace("*", " * ").split()
    const = toks[0] == "const"
    if const:
        toks = toks[1:]
    depth = toks.count("*") + (1 if array else 0)
    return (depth, const and depth > 0, C_KIND[toks[0]])


def c_functions():
    text = open(os.path.join(ROOT, "include", "bn254_verify.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    out = {}
    for ret, name, args in re.findall(r"^\s*((?:const\s+)?\w+(?:\s*\*)?)\s+(bn254_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.M | re.S):
        params = []
        args = " ".join(args.split())
        if args != "void":
            for a in args.split(","):
                m = re.match(r"^(.*?)(\w+)\s*(\[[^\]]*\])?$", a.strip())
                params.append(_c_class(m.group(1), m.group(3) is not None))
        out[name] = (_c_class(ret, False), params)
    return out, text


def _r_class(ty):
    ty = ty.strip()
    depth, const = 0, False
    while ty.startswith("*"):
        m = re.match(r"^\*(const|mut)\s+(.*)$", ty)
        depth += 1
        const = m.group(1) == "const"      # constness of the innermost pointer level is what the C side declares
        ty = m.group(2).strip()
    return (depth, const and depth > 0, R_KIND[ty])


def rust_functions():
    text = open(os.path.join(ROOT, "rust", "bn254-verify-amd-sys", "src", "lib.rs")).read()
    block = re.search(r'extern "C" \{(.*?)\n\}', text, flags=re.S).group(1)
    out = {}
    for name, args, ret in re.findall(r"pub fn (bn254_\w+)\((.*?)\)(?:\s*->\s*([^;]+))?;", block):
        params = [_r_class(a.split(":", 1)[1]) for a in args.split(",") if a.strip()]
        out[name] = (_r_class(ret or "()"), params)
    return out, text


def test_every_c_entry_point_has_the_same_rust_signature():
    c, _ = c_functions()
    r, _ = rust_functions()
    public = {k: v for k, v in c.items() if not k.startswith("bn254_dbg_")}       # the test probes are not part of the binding
    assert len(public) >= 40
    assert set(public) == set(r), (sorted(set(public) - set(r)), sorted(set(r) - set(public)))
    for name, (ret, params) in public.items():
        rret, rparams = r[name]
        assert ret == rret, (name, ret, rret)
        assert len(params) == len(rparams), name
        for i, (a, b) in enumerate(zip(params, rparams)):
            assert a == b, (name, i, a, b)


def test_constants_agree():
    _, ctext = c_functions()
    _, rtext = rust_functions()
    cvals = {}
    for body in re.findall(r"enum\s*\{(.*?)\}", ctext, flags=re.S):
        for name, v in re.findall(r"(BN254_[A-Z0-9_]+)\s*=\s*(-?\d+)u?", body):
            cvals[name] = int(v)
    for name, v in re.findall(r"#define\s+(BN254_[A-Z0-9_]+)\s+(\d+)\b", ctext):
        cvals[name] = int(v)
    rvals = {n: int(v) for n, v in re.findall(r"pub const (BN254_[A-Z0-9_]+): \w+ = (-?\d+);", rtext)}
    assert len(cvals) >= 20 and cvals == rvals


def test_crates_are_complete_files():
    """Cargo manifests, the build script that links the library, the safe wrapper with the reference's surface and its #[cfg(test)] module on tests/golden."""
    for p in ("rust/README.md", "rust/bn254-verify-amd-sys/Cargo.toml", "rust/bn254-verify-amd-sys/build.rs", "rust/bn254-verify-amd/Cargo.toml", "rust/bn254-verify-amd/src/lib.rs"):
        assert os.path.getsize(os.path.join(ROOT, p)) > 200, p
    w = open(os.path.join(ROOT, "rust", "bn254-verify-amd", "src", "lib.rs")).read()
    for needle in ("impl Groth16Verifier", "pub fn verify(", "pub fn verify_batch(", "impl PlonkVerifier", "#[cfg(test)]", "tests/golden", "PrepareInputsFailed", "OpeningPolyMismatch"):
        assert needle in w, needle
    # every sys function the wrapper calls exists in the sys crate
    r, _ = rust_functions()
    for name in set(re.findall(r"sys::(bn254_\w+)\(", w)):
        assert name in r, name
    assert "rustc-link-lib=dylib=bn254_verify_amd" in open(os.path.join(ROOT, "rust", "bn254-verify-amd-sys", "build.rs")).read()
