"""Builds exblas_amd/lib/libexblas.so (HIP, gfx950) in-tree with hipcc.  No torch involved.

Staleness is decided by a content hash of the sources and flags (lib/build.stamp), not by mtimes: the
library travels to the GPU box inside a snapshot whose copy order says nothing about build order.  A rebuild
holds an exclusive file lock and publishes the .so with an atomic rename, so the N ranks of a multi-GPU
launch can all call build() at once.
"""
import fcntl
import hashlib
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
INC = os.path.join(os.path.dirname(PKG), "include")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libexblas.so")
STAMP = os.path.join(LIBDIR, "build.stamp")
SOURCES = ["blas1.hip", "blas2.hip", "trsv.hip", "blas3.hip", "blas3_mfma.hip", "blas3_i8.hip", "blas3_crt.hip", "capi.hip", "comm.hip",
           "generators.cpp"]
# -ffp-contract=off is mandatory: TwoSum/TwoProd must not be fused or re-associated.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function"]


def hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def source_hash():
    h = hashlib.sha256()
    h.update(" ".join(FLAGS).encode())
    for d in (CSRC, INC):
        for f in sorted(os.listdir(d)):
            p = os.path.join(d, f)
            if os.path.isfile(p):
                h.update(f.encode())
                with open(p, "rb") as fh:
                    h.update(fh.read())
    return h.hexdigest()


def stale():
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    try:
        with open(STAMP) as fh:
            return fh.read().strip() != source_hash()
    except OSError:
        return True


def build(force=False, verbose=False, extra_flags=()):
    if not force and not stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not stale():  # another process built it while we waited
            return LIB
        tag = str(os.getpid())
        objs, procs = [], []
        for src in SOURCES:
            obj = os.path.join(LIBDIR, f"{src.replace('.', '_')}.{tag}.o")
            objs.append(obj)
            cmd = [hipcc(), *FLAGS, *extra_flags, "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        failed = False
        for src, p in procs:
            out, _ = p.communicate()
            if p.returncode != 0:
                failed = True
                sys.stderr.write(f"--- {src} ---\n{out}\n")
            elif verbose and out.strip():
                print(out)
        try:
            if failed:
                raise RuntimeError("hipcc failed")
            tmp = LIB + f".{tag}.tmp"
            subprocess.run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp, *objs], check=True)
            os.replace(tmp, LIB)
            with open(STAMP, "w") as fh:
                fh.write(source_hash())
        finally:
            for o in objs:
                if os.path.exists(o):
                    os.remove(o)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
