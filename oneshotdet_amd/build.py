"""Build liboneshotdet_hip.so (gfx950) in-tree with hipcc.  `python -m oneshotdet_amd.build`.

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels to the GPU box with the snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
# OSD_BUILD_TAG=<name> + OSD_BUILD_FLAGS="-D..." build a second, diagnostic copy (lib/liboneshotdet_hip_<name>.so, own
# object dir) for A/B timing through OSD_LIB_PATH; the default build is untouched
TAG = os.environ.get("OSD_BUILD_TAG", "")
LIB = os.path.join(LIBDIR, "liboneshotdet_hip%s.so" % ("_" + TAG if TAG else ""))
SOURCES = ["osd_error.hip", "conv_igemm.hip", "conv_igemm_dma.hip", "conv_igemm_sp.hip", "conv_pred.hip", "conv_px.hip", "conv_wgrad.hip", "conv_wgrad_sk.hip", "backward.hip", "groupnorm_onepass.hip", "loss.hip", "elementwise.hip", "proposals.hip", "box_head.hip", "transforms.hip", "box_train.hip", "evaluation.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result"] + os.environ.get("OSD_BUILD_FLAGS", "").split()


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj" + ("_" + TAG if TAG else ""))
    os.makedirs(objdir, exist_ok=True)
    # every header under csrc/ is a dependency of every object (a handful of small files: a finer map is not worth a stale build)
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join(os.path.dirname(HERE), "include", "oneshotdet_hip.h")]
    # objects of sources that are no longer built (retired kernels) do not stay behind: nothing links them, but they travel with
    # the snapshot and read as live code
    keep = set(src.replace(".hip", ".o") for src in SOURCES)
    for f in os.listdir(objdir):
        if f.endswith(".o") and f not in keep:
            os.remove(os.path.join(objdir, f))
    objs, procs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [HIPCC] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError("hipcc failed on %s" % src)
    if force or procs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
