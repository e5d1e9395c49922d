"""Sanitizer job: the HOST half of every translation unit of libpmf_hip.so under
AddressSanitizer + UndefinedBehaviorSanitizer, driven through the entry points that run without a GPU.

CPU-only by construction (device code is compiled as usual; GPU sanitizers are not available on the GPU
pool): this file is listed in .gpurunignore and never travels to the GPU box."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- sanitizer job: the HOST side of every translation unit under ASan + UBSan -------
SAN_DRIVER = r'''
import ctypes as C, re, sys
lib = C.CDLL(sys.argv[1])
text = re.sub(r"/\*.*?\*/", "", open(sys.argv[2]).read(), flags=re.S)   # (incl. the PMF_TEST_TRANSPORT prototypes: this build has them)
protos = re.findall(r"\b(?:int|const char \*)\s*(pmf_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text)
assert len(protos) > 50, len(protos)
lib.pmf_last_error.restype = C.c_char_p
# 1. no GPU: creation fails with an error code and a message, the out pointer stays NULL
h = C.c_void_p(1234)
rc = lib.pmf_ctx_create(0, C.c_int64(10), C.c_int64(10), 8, 0, C.byref(h))
assert rc < 0 and not h.value and lib.pmf_last_error(), (rc, h.value)
assert lib.pmf_ctx_create(0, C.c_int64(10), C.c_int64(10), 8, 0, None) == -1          # null out pointer
assert lib.pmf_ctx_create(0, C.c_int64(0), C.c_int64(10), 8, 0, C.byref(h)) == -1     # bad dimensions
assert lib.pmf_ctx_create(0, C.c_int64(10), C.c_int64(10), 0, 0, C.byref(h)) == -4    # bad n_factors
assert lib.pmf_ctx_create(0, C.c_int64(10), C.c_int64(10), 8, 7, C.byref(h)) == -1    # bad dtype
n = C.c_int(5)
assert lib.pmf_device_count(C.byref(n)) < 0 and n.value == 0
assert lib.pmf_device_count(None) == -1
# 2. destroy of a context that never existed, twice
assert lib.pmf_ctx_destroy(None) == 0 and lib.pmf_ctx_destroy(None) == 0
# 3. every entry point with a NULL context (and zeros / NULLs for the rest): an error code, never a crash
called = 0
for name, args in protos:
    if name in ("pmf_abi_version", "pmf_last_error", "pmf_device_count", "pmf_ctx_create", "pmf_ctx_destroy"):
        continue
    argv = []
    for a in [x.strip() for x in args.split(",")]:
        if a == "void":
            continue
        if "*" in a:
            argv.append(None)
        elif a.startswith("double"):
            argv.append(C.c_double(0.0))
        elif a.startswith("int64_t"):
            argv.append(C.c_int64(0))
        else:
            argv.append(C.c_int(0))
    fn = getattr(lib, name)
    fn.restype = C.c_int
    rc = fn(*argv)
    assert rc < 0, (name, rc)
    assert lib.pmf_last_error(), name
    called += 1
assert called >= 45, called
print("sanitized entry points exercised:", called)
'''


def test_host_side_under_address_and_ub_sanitizers(tmp_path):
    """Builds the host half of libpmf_hip.so with -fsanitize=address,undefined (device code is compiled as
    usual; GPU sanitizers are not available) and drives what can run without a GPU: the failing
    pmf_ctx_create, null arguments on every entry point, destroy of a null context."""
    import glob
    import __graft_entry__ as g
    hipcc = "/opt/rocm/bin/hipcc"
    rt = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    if not rt:
        pytest.skip("no ASan runtime in this toolchain")
    procs, objs = [], []
    for src in g.SOURCES:
        o = str(tmp_path / src.replace(".hip", ".o"))
        objs.append(o)
        procs.append(subprocess.Popen(
            [hipcc, "-O1", "-g", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-Wno-unused-result"] +
            g.EXTRA_FLAGS.get(src, []) + [
             "-DPMF_TEST_TRANSPORT",     # the hostshm rehearsal transport is host code: it is sanitised with the rest
             "-Xarch_host", "-fsanitize=address", "-Xarch_host", "-fsanitize=undefined",
             "-Xarch_host", "-fno-sanitize-recover=undefined", "-Xarch_host", "-fno-omit-frame-pointer",
             "-I", os.path.join(ROOT, "include"), "-I", g.CSRC, "-c", os.path.join(g.CSRC, src), "-o", o],
            stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate()
        assert p.returncode == 0, out.decode(errors="replace")[-3000:]
    lib = str(tmp_path / "libpmf_hip_san.so")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address", "-fsanitize=undefined",
                        "-shared-libsan", "-o", lib] + objs + ["-L/opt/rocm/lib", "-lrccl"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "__asan_init" in subprocess.run(["nm", "-D", lib], capture_output=True, text=True).stdout
    drv = tmp_path / "drive.py"
    drv.write_text(SAN_DRIVER)
    env = dict(os.environ, LD_PRELOAD=rt[0], ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    out = subprocess.run([sys.executable, str(drv), lib, os.path.join(ROOT, "include", "pmf_hip.h")], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    assert "sanitized entry points exercised" in out.stdout
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr
