"""Builds rtabmap_amd/liblcd_hip.so (the C-ABI of include/lcd.h) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this also runs in the authoring container ("does it build").
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "liblcd_hip.so")
SOURCES = ["knn2_kernels.hip", "knn_mfma_kernels.hip", "resolve_kernels.hip", "tfidf.hip", "bayes.hip", "engine.hip"]
# -ffp-contract=off: the L2 distance must round every product and sum like the reference (no FMA contraction)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
         # MFMA results straight into VGPRs: the top-3 epilogue of knn_mfma_filter_kernel reads them with VALU, and an
         # accumulator in AGPRs costs one v_accvgpr_read per score (f32 MFMA and VALU time are additive on this chip)
         "-mllvm", "-amdgpu-mfma-vgpr-form=1"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: liblcd_hip.so cannot be built (there is no CPU fallback)")


def _stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "lcd.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every HIP translation unit and link the shared library.  Returns the path.
    Several processes may get here at once (one rank per GPU): the build is serialised by a file lock and the library is
    moved into place atomically, so a concurrent loader never sees a half-written file."""
    if not force and not _stale():
        return OUT
    import fcntl
    with open(os.path.join(HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():       # another process built it while this one waited
                return OUT
            return _build_locked(verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(verbose):
    hipcc = _hipcc()
    extra = os.environ.get("LCD_EXTRA_HIPCC_FLAGS", "").split()     # timing experiments only (e.g. -DLCD_MFMA_ABLATE=1)
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    objs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc] + FLAGS + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out))
        if verbose and out.strip():
            print(out, file=sys.stderr)
    tmp = OUT + ".tmp.%d" % os.getpid()
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout)
    os.replace(tmp, OUT)
    return OUT


HOST_OUT = os.path.join(HERE, "liblcd_host.so")
HOST_SOURCES = ["VWDictionaryHip.cpp", "MemoryHip.cpp", "BayesFilterHip.cpp", "RtabmapHip.cpp", "DbLoaderHip.cpp", "c_shim.cpp"]


def build_host(force=False, verbose=False):
    """The C++ host mirror of the reference's VWDictionary / Memory interface (rtabmap_amd/host), linked against the C-ABI."""
    lib = build(force=force, verbose=verbose)
    hdir = os.path.join(HERE, "host")
    if not force and os.path.exists(HOST_OUT):
        t = os.path.getmtime(HOST_OUT)
        deps = [os.path.join(hdir, f) for f in os.listdir(hdir)] + [lib]
        if all(os.path.getmtime(d) <= t for d in deps):
            return HOST_OUT
    cxx = shutil.which("g++") or "g++"
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", HOST_OUT] + [os.path.join(hdir, s) for s in HOST_SOURCES] + \
          ["-L" + HERE, "-llcd_hip", "-ldl", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("host library build failed:\n" + r.stdout)
    return HOST_OUT


SHARD_OUT = os.path.join(HERE, "liblcd_shard.so")


def build_shard(force=False, verbose=False):
    """The multi-GPU driver of include/lcd_shard.h: C++ host code over the C-ABI + RCCL (rtabmap_amd/host/ShardedLcd.cpp)."""
    lib = build(force=force, verbose=verbose)
    src = os.path.join(HERE, "host", "ShardedLcd.cpp")
    if not force and os.path.exists(SHARD_OUT) and os.path.getmtime(SHARD_OUT) >= max(os.path.getmtime(src), os.path.getmtime(lib)):
        return SHARD_OUT
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cxx = shutil.which("g++") or "g++"
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(rocm, "include"), "-o", SHARD_OUT, src,
           "-L" + HERE, "-llcd_hip", "-L" + os.path.join(rocm, "lib"), "-lrccl", "-lamdhip64", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(rocm, "lib")]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("shard library build failed:\n" + r.stdout)
    return SHARD_OUT


P2P_OUT = os.path.join(HERE, "liblcd_p2p.so")


def build_p2p(force=False, verbose=False):
    """The one-shot peer-to-peer exchanges of include/lcd_p2p.h (rtabmap_amd/csrc/p2p_exchange.hip): kernels + host code, HIP only."""
    src = os.path.join(CSRC, "p2p_exchange.hip")
    deps = [src, os.path.join(HERE, "..", "include", "lcd_p2p.h"), os.path.join(HERE, "..", "include", "lcd_shard.h")]
    if not force and os.path.exists(P2P_OUT) and os.path.getmtime(P2P_OUT) >= max(os.path.getmtime(d) for d in deps):
        return P2P_OUT
    tmp = P2P_OUT + ".tmp.%d" % os.getpid()
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-shared", "-o", tmp, src]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("p2p library build failed:\n" + r.stdout)
    os.replace(tmp, P2P_OUT)
    return P2P_OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_host(force="--force" in sys.argv, verbose=True))
    print(build_shard(force="--force" in sys.argv, verbose=True))
    print(build_p2p(force="--force" in sys.argv, verbose=True))
