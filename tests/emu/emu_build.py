"""TEST-ONLY: build the kernel sources for the host with clang++ against tests/emu/shim (the HIP
emulation header) -> tests/emu/_build/libdtqn_emu.so.  Never used by dtqn_amd."""
import hashlib
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(REPO, "dtqn_amd", "csrc")
OUT = os.path.join(HERE, "_build")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def sources():
    return [os.path.join(HERE, "hipemu.cpp")] + sorted(
        os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def build(verbose=False) -> str:
    os.makedirs(OUT, exist_ok=True)
    srcs = sources()
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp"))]
    deps += [os.path.join(REPO, "include", "dtqn_hip.h"), os.path.join(HERE, "shim", "hip", "hip_runtime.h")]
    h = hashlib.sha1()
    extra = os.environ.get("DTQN_EMU_DEFS", "").split()        # e.g. "-DDTQN_OPT=0": an option subset of the kernels (dtqn_device.hpp) on the emulation
    h.update(" ".join(extra).encode())
    for d in sorted(deps):
        h.update(open(d, "rb").read())
    tag = h.hexdigest()[:16]
    lib = os.path.join(OUT, f"libdtqn_emu_{tag}.so")
    if os.path.exists(lib):
        return lib
    if not os.path.exists(CLANG):
        raise RuntimeError("host clang++ not found")
    # one builder at a time (pytest-xdist workers reach this together after a source change and would compile into the same object
    # files); whoever gets the lock second finds the library built
    import fcntl
    with open(os.path.join(OUT, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if os.path.exists(lib):
            return lib
        return _build_locked(srcs, lib, verbose)


def _build_locked(srcs, lib, verbose):
    extra = os.environ.get("DTQN_EMU_DEFS", "").split()
    for f in os.listdir(OUT):
        if f.startswith("libdtqn_emu_") and not extra:     # (option-subset builds live beside the default one)
            os.remove(os.path.join(OUT, f))
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(OUT, os.path.basename(s) + (".x" if extra else "") + ".o")
        cmd = [CLANG, "-x", "c++", "-std=c++17", "-O2", "-fPIC", "-c", "-I" + os.path.join(HERE, "shim"),
               "-I" + os.path.join(REPO, "include"), "-I" + CSRC, "-Wno-unused-value", s, "-o", o] + extra
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(o)
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"emu compile failed for {s}:\n{out.decode()}")
        if verbose and out:
            print(out.decode())
    subprocess.check_call([CLANG, "-shared", "-o", lib + ".tmp"] + objs + ["-lpthread"])
    os.replace(lib + ".tmp", lib)           # the library appears complete or not at all
    for o in objs:
        os.remove(o)
    return lib


if __name__ == "__main__":
    print(build(verbose=True))
