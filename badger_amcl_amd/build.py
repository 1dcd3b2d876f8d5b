"""Builds libbadger_pf_hip.so (gfx950) in-tree with hipcc.  No CPU fallback exists."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "engine.hip")
OUT = os.path.join(HERE, "libbadger_pf_hip.so")
# the RCCL collectives of the sharded path: a separate object linked against librccl (570 MB), loaded by
# bpf_shard_bootstrap only on ranks that cannot use the mailbox exchange
SRC_RCCL = os.path.join(HERE, "csrc", "collectives_rccl.cpp")
OUT_RCCL = os.path.join(HERE, "libbadger_pf_rccl.so")


def deps():
    """Every source the one translation unit pulls in: csrc/* and the C-ABI header."""
    csrc = os.path.join(HERE, "csrc")
    return [os.path.join(csrc, f) for f in sorted(os.listdir(csrc))] + \
        [os.path.join(os.path.dirname(HERE), "include", "badger_pf.h")]


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP engine cannot be built")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in deps())


def build_rccl(force=False, verbose=False):
    if not force and os.path.exists(OUT_RCCL) and os.path.getmtime(OUT_RCCL) >= os.path.getmtime(SRC_RCCL):
        return OUT_RCCL
    rocm_lib = os.path.join(os.path.dirname(os.path.dirname(hipcc())), "lib")
    cmd = [hipcc(), "-O2", "-std=c++17", "-fPIC", "-shared", "-o", OUT_RCCL, SRC_RCCL, "-L", rocm_lib, "-lrccl",
           "-Wl,-rpath," + rocm_lib]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return OUT_RCCL


def build(force=False, verbose=False):
    build_rccl(force, verbose)
    if not force and not needs_build():
        return OUT
    # BPF_EXTRA_FLAGS: experiment builds (e.g. "-DBPF_FIELD_UNROLL=4 -DBPF_FIELD_WAVES=5"); not used by the product
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"] + \
        os.environ.get("BPF_EXTRA_FLAGS", "").split() + ["-o", OUT, SRC]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
