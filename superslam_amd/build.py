"""Build libsuperslam_hip.so (hipcc, gfx950 only) in-tree: superslam_amd/lib/libsuperslam_hip.so."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libsuperslam_hip.so")
# The shipped library: ONE kernel per layer, no run-time kernel selection (include/sship.h, "Environment").
SOURCES = ["api.hip", "sp_kernels.hip", "sp_convs.hip", "conv_pp.hip", "conv_pp128.hip", "conv_fuse2.hip", "lg_kernels.hip", "ep_kernels.hip", "probe.hip", "shard_rccl.hip"]
# Developer build (lib/variants/dev.so, -DSSHIP_DEV_SWITCHES=1): the same sources with the A/B switches and phase traces compiled in, plus
# the rejected kernels they select: the lock-step strip conv (r01), Winograd conv2a/2b (r04: -25 %), the 16-wave FFN (r04: +-0), the
# LDS-resident-key attention (r05: -8 %).  tests/test_gpu_alt_paths.py and scripts/dev/* load it explicitly (superslam_amd._lib.set_library_path via scripts/_devlib.py).
DEV_SOURCES = SOURCES + ["conv_strip.hip", "conv_wino.hip", "lg_ffn16.hip", "lg_attn_res.hip"]
DEV_FLAGS = ["-DSSHIP_DEV_SWITCHES=1"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable"]
# per-file flags.  -fno-honor-nans: under IEEE NaN semantics every fmaxf() operand that comes out of an MFMA is
# canonicalised first (a second v_max_f32 per value) - the attention softmax's running max and every ReLU / max-pool
# of the conv epilogues paid for that.  None of these kernels produces or tests for a NaN (masked scores are -inf,
# never inf - inf); infinities keep their meaning.
FILE_FLAGS = {f: ["-fno-honor-nans"] for f in ("lg_kernels.hip", "lg_attn_res.hip", "lg_ffn16.hip", "conv_wino.hip", "conv_pp.hip", "conv_pp128.hip", "conv_strip.hip", "sp_convs.hip")}
# LightGlue kernels: no packed-fp32 VALU (v_pk_mul_f32, v_pk_fma_f32, ...).  Two workgroups share a CU there so that one's
# GELU / LayerNorm / softmax VALU rides next to the other's MFMA stream, and next to an MFMA stream a packed-fp32
# instruction costs 17 clocks per wave-instruction against 6.4 for a plain one (profiles/r02_valu_rates_next_to_mfma.txt):
# two scalar instructions beat one packed one.  -1 % on a 64-pair LightGlue call.
# (lg_ffn16.hip: its 16 waves move through the phases together, nothing issues MFMAs next to its GELU phase - packed fp32 was tried there
# (GELU 9.5 k -> 7.9 k clocks per chunk) but the register pairs it needs pushed the rotary epilogue into scratch: profiles/r04_c_*)
for _f in ("lg_kernels.hip", "lg_attn_res.hip", "lg_ffn16.hip", "conv_wino.hip"):
    FILE_FLAGS[_f] = FILE_FLAGS[_f] + ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(os.path.join(LIBDIR, "obj"), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "sship.h"))
    hdr_time = _newest(headers)
    jobs = []
    objs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(LIBDIR, "obj", src.replace(".hip", ".o"))
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(sp), hdr_time):
            jobs.append([hipcc, *FLAGS, *FILE_FLAGS.get(src, []), "-c", sp, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl"])
    return LIB


def build_dev(force: bool = False) -> str:
    """lib/variants/dev.so, rebuilt when a source is newer (the A/B tests and scripts load it)."""
    out = os.path.join(LIBDIR, "variants", "dev.so")
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(os.path.dirname(HERE), "include", "sship.h"), os.path.abspath(__file__)]
    if force or not os.path.exists(out) or os.path.getmtime(out) < _newest(deps):
        build_variant("dev", DEV_FLAGS)
    return out


def build_variant(name: str, extra_flags) -> str:
    """Developer A/B builds: same sources + extra -D flags -> lib/variants/<name>.so (load with superslam_amd._lib.set_library_path / bench.py --library / SSHIP_DEV_LIBRARY in the dev scripts)."""
    vdir = os.path.join(LIBDIR, "variants")
    odir = os.path.join(vdir, "obj_" + name)
    os.makedirs(odir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []

    def one(src):
        obj = os.path.join(odir, src.replace(".hip", ".o"))
        r = subprocess.run([hipcc, *FLAGS, *FILE_FLAGS.get(src, []), *extra_flags, "-c", os.path.join(CSRC, src), "-o", obj],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stdout + r.stderr)
        return obj

    srcs = DEV_SOURCES if "-DSSHIP_DEV_SWITCHES=1" in extra_flags else SOURCES
    with ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(one, srcs))
    out = os.path.join(vdir, name + ".so")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs, "-ldl"], check=True)
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--dev":
        print(build_dev(force="--force" in sys.argv))
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":
        print(build_variant(sys.argv[2], sys.argv[3:]))
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose=True))
