"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/kt_engine.h declares, and refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from kube_throttler_amd import engine as E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    E.build()
    return C.CDLL(E.LIB_PATH)


def declared_symbols():
    with open(os.path.join(ROOT, "include", "kt_engine.h")) as fh:
        text = fh.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kt_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported(built_lib):
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(built_lib, s), f"{s} declared in include/kt_engine.h but not exported"
    assert set(E.EXPORTS) == set(syms)


def test_version_string(built_lib):
    built_lib.kt_version.restype = C.c_char_p
    assert b"gfx950" in built_lib.kt_version()


def test_no_cpu_fallback():
    """On a box without a GPU engine creation must fail loudly, not fall back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(E.EngineError) as ei:
        E.Engine(4, 4, 16, 4, 2)
    assert ei.value.code == -6


def test_invalid_config_rejected(built_lib):
    cfg = E.KtConfig(0, 4, 16, 4, 2, -1, 0)
    h = C.c_void_p()
    assert E.lib().kt_engine_create(C.byref(cfg), C.byref(h)) == -1


def test_product_never_reaches_for_the_oracle():
    """oracle/ is test infrastructure: nothing under kube_throttler_amd/ may import, load or link it, bench.py only in
    its cpu_baseline leg and in the checker of `--verify` (verify_against_oracle: the run's result held against the oracle AFTER
    the timed region, never measured), __graft_entry__ only in build() (compiling the checker) and smoke() (checking)."""
    import ast
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pat = re.compile(r"(^\s*(from|import)\s+oracle\b)|kt_oracle|libkt_oracle|oracle/", re.M)
    for dirpath, _, files in os.walk(os.path.join(root, "kube_throttler_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip", ".c")) or f == "Makefile":
                with open(os.path.join(dirpath, f), errors="replace") as fh:
                    m = pat.search(fh.read())
                assert m is None, f"{os.path.join(dirpath, f)} refers to the oracle: {m.group(0)!r}"
    # bench.py: every import of the oracle sits inside the `if ... not args.no_cpu_baseline` block
    src = open(os.path.join(root, "bench.py")).read()
    tree = ast.parse(src)
    imports = [n for n in ast.walk(tree) if isinstance(n, ast.ImportFrom) and (n.module or "").split(".")[0] == "oracle"]
    assert len(imports) == 2
    guard = next(n for n in ast.walk(tree) if isinstance(n, ast.If) and "no_cpu_baseline" in ast.unparse(n.test))
    checker = next(n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == "verify_against_oracle")
    inside = lambda node, imp: node.lineno < imp.lineno <= node.end_lineno
    assert sorted((inside(guard, i), inside(checker, i)) for i in imports) == [(False, True), (True, False)]
    # ... and the checker is only ever called under `if args.verify`
    calls = [n for n in ast.walk(tree) if isinstance(n, ast.Call) and getattr(n.func, "id", "") == "verify_against_oracle"]
    verify_ifs = [n for n in ast.walk(tree) if isinstance(n, ast.If) and "args.verify" in ast.unparse(n.test)]
    assert calls and all(any(inside(g, c) for g in verify_ifs) for c in calls)
    assert not [n for n in ast.walk(tree) if isinstance(n, ast.Import) and any(a.name.startswith("oracle") for a in n.names)]


def test_headers_are_plain_c(tmp_path):
    """The boundary is a C ABI: include/*.h must compile as C99 on their own (cgo compiles them with a C compiler),
    and every size the binding relies on is fixed-width."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "abi.c"
    src.write_text('#include "kt_engine.h"\n#include "kt_snapshot.h"\n'
                   'int main(void) { kt_config c; kt_snapshot s; kt_status st; kt_amounts a;\n'
                   '  (void)c; (void)s; (void)st; (void)a;\n'
                   '  return (sizeof(c.pod_capacity) == 8 && sizeof(s.n_pods) == 8 && KT_OK == 0) ? 0 : 1; }\n')
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", f"-I{root}/include", str(src), "-o", str(exe)])
    assert subprocess.call([str(exe)]) == 0


def test_native_latency_shim_builds_against_the_header(tmp_path, built_lib):
    """tools/native/busy_latency.c (the single-pod calls timed from native threads, what tools/latency_bench.py loads on
    the GPU box) is plain C over include/kt_engine.h: it must compile without a warning and link against the library."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "kube_throttler_amd", "csrc")
    so = tmp_path / "busy.so"
    subprocess.check_call(["gcc", "-std=gnu99", "-O2", "-Wall", "-Wextra", "-Werror", "-shared", "-fPIC", f"-I{root}/include",
                           os.path.join(root, "tools", "native", "busy_latency.c"), f"-L{csrc}", "-lkt_engine", "-lpthread",
                           f"-Wl,-rpath,{csrc}", "-o", str(so)])
    import ctypes
    lib = ctypes.CDLL(str(so))
    assert hasattr(lib, "kt_native_check_latency")


def test_partial_layout_query_needs_no_gpu(built_lib):
    """kt_partial_layout: what a host's own collective (and tests/partial_layout.py) lays the partial buffer out by — the
    stride and offsets the kernels are compiled against, for every dimension count the engine takes."""
    for D in range(1, 17):
        lo = E.partial_layout(D)
        assert lo["stride"] >= 2 * D + 2
        spans = sorted([(lo["values"], D), (lo["presence"], D), (lo["pods"], 1), (lo["errors"], 1)])
        end = 0
        for off, n in spans:  # the four parts do not overlap and lie inside the row
            assert off >= end
            end = off + n
        assert end <= lo["stride"]
    with pytest.raises(E.EngineError):
        E.partial_layout(17)
