"""Builds libnhwhip.so (hand-written HIP for gfx950) in-tree with hipcc.  No CPU fallback exists."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libnhwhip.so")
SOURCES = ["nhw_front.hip", "nhw_tail.hip", "nhw_low.hip", "nhw_api.hip", "nhw_dec.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-value"]
if os.environ.get("NHW_DEV"):
    FLAGS.append("-DNHW_DEV")       # phase stamps and pass switches of the front kernels (developer builds only)
if os.environ.get("NHW_EXTRA_FLAGS"):
    FLAGS += os.environ["NHW_EXTRA_FLAGS"].split()   # developer experiments (e.g. -DLW=64: tools/dev/build_variant.sh)
if os.environ.get("NHW_PROFILE"):
    FLAGS.append("-DNHW_PROFILE")   # in-kernel per-pass timers (developer builds only)


def _stale() -> bool:
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(os.path.dirname(HERE), "include", "nhw_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False) -> str:
    if not force and not _stale():
        return SO
    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc"
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(obj)
        procs.append(subprocess.Popen([hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]))
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", SO])
    return SO


if __name__ == "__main__":
    print(build(force=True))
