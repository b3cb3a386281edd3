"""g++ builds of the C++ test / example binaries above the C ABI.

Test artefacts go to tests/_build/ (git-ignored; they travel to the GPU box with the snapshot) - never into superslam_amd/lib/,
which is what a wheel of the package would ship.  `sanitize=True` is the reference's SUPERSLAM_SANITIZE
build (CMakeLists.txt:81-90: -fsanitize=address,undefined -fno-omit-frame-pointer) for the host shim; CPU only - GPU
AddressSanitizer is not available on this pool."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "superslam_amd", "lib")
OUTDIR = os.path.join(ROOT, "tests", "_build")
SAN_FLAGS = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined", "-g"]


def cpp_binary(name, srcs, deps=(), includes=(), link_lib=True, extra=(), opt="-O1", sanitize=False, force=False, outdir=None):
    """Compile `srcs` into <outdir>/<name> when any of srcs + deps is newer; returns the path."""
    outdir = outdir or OUTDIR
    os.makedirs(outdir, exist_ok=True)
    out = os.path.join(outdir, name + ("_san" if sanitize else ""))
    newest = max(os.path.getmtime(p) for p in [*srcs, *deps, os.path.abspath(__file__)])
    if force or not os.path.exists(out) or os.path.getmtime(out) < newest:
        cmd = ["g++", "-std=c++17", opt, "-Wall", *(SAN_FLAGS if sanitize else []), *["-I" + i for i in includes],
               "-I" + os.path.join(ROOT, "include"), *srcs, "-o", out]
        if link_lib:
            cmd += ["-L" + LIBDIR, "-lsuperslam_hip", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"]
        cmd += list(extra)
        subprocess.check_call(cmd)
    return out


def sanitizer_env():
    """The HIP runtime's own start-up allocations are not ours to judge: leaks off, everything else fatal."""
    return dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1",
                UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
