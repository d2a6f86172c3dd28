"""The Rust shim crate (rust/rodio-hip) has never met a compiler (no cargo in this image), so this is what a compiler and a
linker would check first: every `extern "C"` declaration and `#[repr(C)]` struct of src/ffi.rs is compared, type by type, with
the prototype / struct of the same name in include/rodio_hip.h; ffi.rs covers the whole header; src/lib.rs calls nothing that
ffi.rs does not declare, with the right number of arguments; ffi.rs is what tools/gen_rust_ffi.py generates."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# Rust type -> the C spellings it may stand for (after whitespace / const normalisation below)
SCALARS = {
    "f32": "float", "f64": "double", "i8": "int8_t", "u8": "uint8_t", "i16": "int16_t", "u16": "uint16_t", "i32": "int32_t",
    "u32": "uint32_t", "i64": "int64_t", "u64": "uint64_t", "usize": "size_t", "core::ffi::c_void": "void", "c_void": "void",
    "core::ffi::c_char": "char", "RhStatus": "rh_status", "RhStream": "rh_stream", "RhEvent": "rh_event",
}
STRUCTS = {"RhRlm": "rh_rlm", "RhRlmConfig": "rh_rlm_config", "RhEcho": "rh_echo", "RhResampler": "rh_resampler", "RhLimitParams": "rh_limit_params",
           "RhAgcParams": "rh_agc_params", "RhComm": "rh_comm", "RhWavInfo": "rh_wav_info", "RhRlmGeometryInfo": "rh_rlm_geometry_info", "RhUniformSeg": "rh_uniform_seg", "RhWideSrc": "rh_wide_src"}
CRATE = os.path.join(ROOT, "rust", "rodio-hip")


def rust_to_c(t):
    t = t.strip()
    if t.startswith("*mut "):
        return rust_to_c(t[5:]) + " *"
    if t.startswith("*const "):
        return rust_to_c(t[7:]) + " const *"
    if t in SCALARS:
        return SCALARS[t]
    if t in STRUCTS:
        return STRUCTS[t]
    raise AssertionError(f"Rust type {t!r} has no C counterpart in this table")


def norm_c(t):
    """'const float *const *srcs' style spellings -> canonical 'const float * const *' tokens, name stripped by the caller"""
    toks = re.sub(r"\s+", " ", t.replace("*", " * ")).strip().split(" ")
    if len(toks) >= 2 and toks[0] == "const":  # 'const T' -> 'T const' (east const, what rust_to_c emits)
        toks = [toks[1], "const"] + toks[2:]
    return " ".join(toks)


def c_params(proto):
    inner = proto[proto.index("(") + 1: proto.rindex(")")].strip()
    if inner in ("", "void"):
        return []
    out = []
    for p in inner.split(","):
        p = p.strip()
        m = re.match(r"^(.*?)(\b[A-Za-z_][A-Za-z0-9_]*)?(\[\d*\])?$", p)
        ty, name, arr = m.group(1), m.group(2), m.group(3)
        if name in ("float", "void", "char", "rh_stream", "rh_event") or (name and name.endswith("_t")):  # unnamed parameter
            ty, name = ty + name, None
        ty = norm_c(ty)
        if arr:  # `float out[8]` decays to a pointer
            ty = ty + " *"
        out.append(ty)
    return out


def header():
    text = open(os.path.join(ROOT, "include", "rodio_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    protos = {}
    for m in re.finditer(r"\b([A-Za-z_][A-Za-z0-9_ \*]*?)\b(rh_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        protos[m.group(2)] = (norm_c(m.group(1)), c_params("(" + m.group(3) + ")"))
    structs = {}
    for m in re.finditer(r"typedef struct (rh_[a-z0-9_]+)\s*\{(.*?)\}\s*\1\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            mm = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*(?:\[\d+\])?(?:\s*,\s*[A-Za-z_][A-Za-z0-9_]*(?:\[\d+\])?)*)$", decl, flags=re.S)
            ty, names = norm_c(mm.group(1)), mm.group(2)  # `const float *src` -> type `float const *`, name `src`
            for n in names.split(","):
                n = n.strip()
                a = re.match(r"^(\w+)\[(\d+)\]$", n)
                fields.append((a.group(1), f"{ty}[{a.group(2)}]") if a else (n, ty))
        structs[m.group(1)] = fields
    return protos, structs


def rust_blocks():
    code = open(os.path.join(CRATE, "src", "ffi.rs")).read()
    code = re.sub(r"//[^\n]*", "", code)
    fns = {}
    for blk in re.findall(r'extern "C"\s*\{(.*?)\n\}', code, flags=re.S):
        for m in re.finditer(r"pub fn (rh_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", blk, flags=re.S):
            args = [a.strip() for a in re.sub(r"\s+", " ", m.group(2)).split(",") if a.strip()]
            fns[m.group(1)] = ([a.split(":", 1)[1].strip() for a in args], (m.group(3) or "()").strip())
    structs = {}
    for m in re.finditer(r"#\[repr\(C\)\](?:\s*#\[derive\([^)]*\)\])?\s*pub struct (\w+)\s*\{(.*?)\}", code, flags=re.S):
        fields = []
        for f in m.group(2).split(","):
            f = re.sub(r"\s+", " ", f).strip()
            if not f:
                continue
            name, ty = f.replace("pub ", "").split(":", 1)
            fields.append((name.strip(), ty.strip()))
        structs[m.group(1)] = fields
    return fns, structs


def test_every_rust_extern_matches_the_header_prototype():
    protos, _ = header()
    fns, _ = rust_blocks()
    assert sorted(fns) == sorted(protos), (sorted(set(protos) - set(fns)), sorted(set(fns) - set(protos)))  # the whole header, nothing else
    for name, (args, ret) in sorted(fns.items()):
        assert name in protos, f"{name} is declared in ffi.rs but not in include/rodio_hip.h"
        c_ret, c_args = protos[name]
        assert (ret == "()" and c_ret == "void") or norm_c(rust_to_c(ret)) == c_ret, (name, ret, c_ret)
        assert len(args) == len(c_args), (name, args, c_args)
        for i, (r, c) in enumerate(zip(args, c_args)):
            want = norm_c(rust_to_c(r))
            # top-level const of a by-value parameter / pointer (`T *const p`) is not part of the ABI
            c = re.sub(r" \* const$", " *", c)
            assert want == c, f"{name} argument {i}: Rust `{r}` = C `{want}`, header says `{c}`"


def test_repr_c_structs_match_the_header_field_by_field():
    _, cstructs = header()
    _, rstructs = rust_blocks()
    checked = 0
    for rname, rfields in rstructs.items():
        if rfields and rfields[0][0] == "_private":  # opaque handle
            assert STRUCTS[rname] not in cstructs  # ... and the header keeps it opaque too
            continue
        cname = STRUCTS[rname]
        assert cname in cstructs, (rname, cname)
        cf = cstructs[cname]
        assert [n for n, _ in rfields] == [n for n, _ in cf], (rname, [n for n, _ in rfields], [n for n, _ in cf])
        for (n, rt), (_, ct) in zip(rfields, cf):
            a = re.match(r"^\[(\w+); (\d+)\]$", rt)
            want = f"{SCALARS[a.group(1)]}[{a.group(2)}]" if a else norm_c(rust_to_c(rt))
            assert want == norm_c(ct) or want == ct, (rname, n, rt, ct)
        checked += 1
    assert checked >= 6


def test_the_shim_calls_only_what_ffi_declares():
    """src/lib.rs: every rh_* call names a function of ffi.rs and passes as many arguments as the prototype has."""
    fns, _ = rust_blocks()
    code = open(os.path.join(CRATE, "src", "lib.rs")).read()
    code = re.sub(r"//[^\n]*", "", code)
    used = set()
    for m in re.finditer(r"\b(rh_[a-z0-9_]+)\s*\(", code):
        name = m.group(1)
        assert name in fns, f"lib.rs calls {name}, which ffi.rs does not declare"
        used.add(name)
        depth, i, args, cur = 0, m.end(), 0, False
        while True:  # count top-level commas up to the matching parenthesis
            c = code[i]
            if c in "([{":
                depth += 1
            elif c in ")]}":
                if depth == 0:
                    break
                depth -= 1
            elif c == "," and depth == 0:
                args += 1
            if not c.isspace() and not (c == ")" and depth == 0):
                cur = True
            i += 1
        n = args + 1 if cur else 0
        assert n == len(fns[name][0]), f"lib.rs calls {name} with {n} arguments, the header has {len(fns[name][0])}"
    for must in ("rh_rlm_stream_block_v", "rh_uniform_segments", "rh_uniform_segments_dev", "rh_mix_sum", "rh_biquad", "rh_limit", "rh_agc", "rh_resampler_process", "rh_echo_process"):
        assert must in used, must
    for item in ("pub struct GpuSource<I: Source>", "pub struct GpuMixer", "fn late_join", "fn try_seek", "pub struct SpanReader", "pub struct UniformPlanner", "impl<I: Source> Source for GpuSource<I>",
                 "impl Source for GpuMixer"):
        assert item in code, item


def test_ffi_rs_is_what_the_generator_writes(tmp_path):
    import subprocess
    import sys

    before = open(os.path.join(CRATE, "src", "ffi.rs")).read()
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_ffi.py")], check=True, capture_output=True)
    assert open(os.path.join(CRATE, "src", "ffi.rs")).read() == before, "rust/rodio-hip/src/ffi.rs is stale: run python tools/gen_rust_ffi.py"


def test_crate_files_exist():
    for f in ("Cargo.toml", "build.rs", os.path.join("src", "lib.rs"), os.path.join("src", "ffi.rs")):
        assert os.path.exists(os.path.join(CRATE, f)), f
    assert 'rodio = { version = "0.22"' in open(os.path.join(CRATE, "Cargo.toml")).read()


def _cpp_public_methods(cls):
    """Names of the public member functions of class `cls` in include/rodio_hip.hpp (its own public sections; constructors,
    destructors and operators aside)."""
    text = open(os.path.join(ROOT, "include", "rodio_hip.hpp")).read()
    start = text.index(f"class {cls} ")
    body = text[text.index("{", start) + 1:]
    depth, end = 1, 0
    for i, c in enumerate(body):  # the class body
        depth += c == "{"
        depth -= c == "}"
        if depth == 0:
            end = i
            break
    body = body[:end]
    names, public = set(), False
    for line in body.split("\n"):
        s = line.strip()
        if re.match(r"^(public|protected|private):", s):
            public = s.startswith("public")
            continue
        if not public or not line.startswith("    ") or line.startswith("     "):  # members sit at one indentation level
            continue
        m = re.match(r"^(?:virtual |static |explicit |inline )*[\w:<>,\*& ]+?[ \*&]([a-z_][a-z0-9_]*)\s*\(", s)
        if m and not s.startswith(("//", "///", "using ", "struct ", "class ", "enum ", "return ", "if ", "for ", "throw ")) and m.group(1) not in (cls,):
            names.add(m.group(1))
    return names


def test_the_rust_twin_has_every_public_method_of_the_cpp_host_mirror():
    """the crate is the twin of include/rodio_hip.hpp, method for method.  Every public member function of
    rodio_hip::GpuSource / GpuMixer (and of their base, detail::BlockPump) has a same-named method in src/lib.rs -- with the few
    spellings that MUST differ between the languages listed here, each with its reason."""
    code = open(os.path.join(CRATE, "src", "lib.rs")).read()
    rust_fns = set(re.findall(r"\bfn ([a-z_][a-z0-9_]*)\s*[<(]", code))
    other = {
        "read": "next_sample",  # Source::read (the bulk form of next()): rodio's trait has no such method; the pump serves next() from its block
        "inner": "inner", "into_inner": "into_inner",
        "channels": "channels", "sample_rate": "sample_rate",
    }
    # overloads of the C++ side get names of their own in Rust
    overloads = {"add": ["add", "add_filtered", "add_chain", "add_chain_filtered"]}
    missing = []
    for cls in ("BlockPump", "GpuSource", "GpuMixer"):
        names = _cpp_public_methods(cls)
        assert len(names) >= (4 if cls == "BlockPump" else 8), (cls, names)
        for n in sorted(names):
            want = overloads.get(n, [other.get(n, n)])
            if cls == "BlockPump" and n == "prepare":
                want = ["prepare_stream", "prepare"]
            if cls == "BlockPump" and n == "read_device":
                want = ["read_device_impl", "read_device"]
            for w in want:
                if w not in rust_fns:
                    missing.append(f"{cls}::{n} -> fn {w}")
    assert not missing, missing
    # ... and what the twin needs besides: the structs and options of the mixer, the device hand-off trait, the reaper, the copy stream
    for item in ("pub struct MixerFilter", "pub reference_exact_filters: bool", "pub host_threads: u32", "pub trait DeviceChain", "struct Reaper", "copy_stream: RhStream",
                 "rh_rlm_stream_keep_history", "rh_filter_scan_ok", "std::thread::scope", "pub struct ChainStats", "pub fn with_channels", "unsafe impl Sync for State",
                 "pub tail: usize"):
        assert item in code, item
