"""Build libdiffnet_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

`python -m diffnet_amd.build` or `diffnet_amd.build.build()`.  Cross-compiles without a GPU.
The .so is git-ignored but travels with the gpurun snapshot.
"""
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdiffnet_hip.so")
STAMP = os.path.join(HERE, ".libdiffnet_hip.stamp")
ARCH = "gfx950"
SOURCES = ["dn_api.hip", "probe.hip", "poisson_fused.hip", "poisson2d_q1_cf.hip", "poisson2d_q1_g2.hip", "poisson2d_q1_g3.hip", "poisson2d_q1_g4.hip", "poisson3d_q1_g2.hip", "poisson3d_q1_g3.hip", "poisson3d_q1_g4.hip", "poisson3d_q1_cf.hip", "poisson3d_gen.hip", "gauss_pt_eval.hip", "winding.hip", "fdm.hip", "instnorm_act.hip", "fsdt.hip", "fsdt_st.hip", "upconv_out.hip", "upconv3d_out.hip", "conv3d_wrw.hip", "conv2d_k4s2.hip", "conv2d_k4s2_v2.hip", "conv3d_k4s2.hip", "conv2d_direct.hip"]
HEADERS = ["dn_common.h", "fsdt_common.h", "poisson_elem.h", "poisson_common.h", "poisson2d_q1.inl", "poisson3d_q1.inl", os.path.join("..", "..", "include", "diffnet_hip.h")]
# -amdgpu-sdwa-peephole=0: on gfx950 an SDWA (like a DPP or v_readlane) instruction costs a SIMD ~33 cycles once two or more waves
# share it -- 14 plain VALU instructions -- so the byte-select forms the peephole creates for mask tests are a large net loss
# (tools/micro/valu_mem.hip, profiles/r2_valu_mem.txt)
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-fno-slp-vectorize", f"--offload-arch={ARCH}", "-mllvm", "-amdgpu-sdwa-peephole=0",
         "-Wall", "-Wno-unused-variable", "-Wno-unused-but-set-variable"] + os.environ.get("DN_EXTRA_FLAGS", "").split()


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libdiffnet_hip.so cannot be built")


def _digest():
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read().strip() == dig:
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    objs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out)
            raise RuntimeError(f"hipcc failed on {src}")
        if verbose and out.strip():
            print(out)
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as fh:
        fh.write(dig)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
