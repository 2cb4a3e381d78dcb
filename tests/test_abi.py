"""The C-ABI library loads and exports every symbol include/rt355.h declares (no compute
calls: there is no GPU on the CPU tier)."""
import os
import re
import subprocess

import pytest

from rsraytracingv3_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rt355.h")).read()
    return sorted(set(re.findall(r"RT355_API\s+[\w\s\*]+?\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_boundary():
    names = declared_symbols()
    for must in ("rt_create", "rt_destroy", "rt_set_skybox", "rt_set_random", "rt_set_scene", "rt_resize",
                 "rt_render", "rt_sync", "rt_read_rgba8", "rt_read_rgbf32", "rt_last_error", "rt_get_stats",
                 "rt_set_partition"):
        assert must in names


def test_library_exports_every_declared_symbol():
    lib = _native.load()
    assert lib.rt_abi_version() == 5
    for name in declared_symbols():
        assert hasattr(lib, name), "librt355.so does not export %s" % name


def test_ctypes_table_covers_the_header():
    assert sorted(_native.SIGNATURES) == declared_symbols()


def test_no_cxx_or_torch_types_in_the_exports():
    out = subprocess.run(["nm", "-D", "--defined-only", _native.LIB_PATH], check=True, capture_output=True, text=True).stdout
    exported = [ln.split()[-1] for ln in out.splitlines() if " T " in ln]
    assert sorted(exported) == declared_symbols()  # nothing mangled leaks out (-fvisibility=hidden)


def test_create_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from rsraytracingv3_amd import Device
    with pytest.raises(_native.RtError) as e:
        Device(0)
    assert e.value.code == -2 and "no CPU fallback" in str(e.value)


def test_product_package_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under rsraytracingv3_amd/ may import, link or
    include it."""
    pkg = os.path.join(ROOT, "rsraytracingv3_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "rt_oracle" not in text and "librt_oracle" not in text, f


def test_one_hip_runtime_whatever_the_import_order():
    """librt355.so and PyTorch must share ONE HIP/HSA runtime in a process: with two copies mapped the second
    one finds no GPU.  _native.load() therefore loads torch's bundled runtime first (without importing torch)."""
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from rsraytracingv3_amd import _native\n"
        "%s\n"
        "libs = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l})\n"
        "print(len(libs), libs)\n"
    )
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for order in ("_native.load(); import torch", "import torch; _native.load()"):
        out = subprocess.run([sys.executable, "-c", code % (root, order)], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-1500:]
        assert out.stdout.split()[0] == "1", "%s -> %s" % (order, out.stdout)


def test_rust_binding_sketch_matches_the_header():
    """INTEGRATION.md's `extern "C"` block cannot be compiled here (no Rust toolchain); at least every function it
    declares must exist in include/rt355.h with the same number of parameters."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    header = open(os.path.join(root, "include", "rt355.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    c_params = {}
    for m in re.finditer(r"RT355_API\s+[\w\s\*]+?\b(rt_\w+)\s*\(([^)]*)\)\s*;", header):
        args = m.group(2).strip()
        c_params[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    rust = re.findall(r"fn (rt_\w+)\(([^)]*)\)\s*->", text)
    assert len(rust) >= 15
    for name, args in rust:
        assert name in c_params, "%s is not in the header" % name
        n = 0 if not args.strip() else args.count(":")
        assert n == c_params[name], "%s: %d parameters in the Rust sketch, %d in the header" % (name, n, c_params[name])
