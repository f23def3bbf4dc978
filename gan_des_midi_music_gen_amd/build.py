"""In-tree build of libgdm_hip.so (hipcc, gfx950 only).  ``python -m gan_des_midi_music_gen_amd.build``."""
import concurrent.futures
import glob
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
# GDM_BUILD_TAG=<tag> (experiments only, with GDM_HIPCC_FLAGS): objects and library of a variant build live beside the
# shipped ones (csrc/_obj_<tag>/, libgdm_hip_<tag>.so); a process picks the variant up with GDM_LIB_TAG=<tag>.
TAG = os.environ.get("GDM_BUILD_TAG", "")
OBJ = os.path.join(CSRC, "_obj" + (f"_{TAG}" if TAG else ""))
LIB = os.path.join(PKG, "libgdm_hip" + (f"_{TAG}" if TAG else "") + ".so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         # MFMA results straight into VGPRs: every kernel here post-processes its accumulators with VALU code and none
         # needs more than 256 registers, so the AGPR form only adds v_accvgpr_read/write traffic (5-15 % of the VALU
         # instructions of the issue-bound conv kernels)
         "-mllvm", "-amdgpu-mfma-vgpr-form=1",
         # no NaN handling in the arithmetic: without it every MFMA result that reaches an fmaxf is first
         # canonicalised (v_max_f32 x, x, x) -- 16 extra VALU per 16 pooled pixels in the issue-bound conv epilogues
         "-fno-honor-nans"]
# per-source additions to FLAGS (experiments: GDM_HIPCC_FILE_FLAGS="gemm_bf16.hip:-mllvm,-amdgpu-sched-strategy=max-ilp")
PER_FILE_FLAGS = {}
for _spec in os.environ.get("GDM_HIPCC_FILE_FLAGS", "").split():
    _name, _, _fl = _spec.partition(":")
    PER_FILE_FLAGS[_name] = _fl.split(",")
EXPERIMENT = os.environ.get("GDM_HIPCC_FLAGS", "").split()     # experiment switches (-D...), empty for the shipped build
if EXPERIMENT:
    # an instrumented / variant build says so: gdm_build_flavor() returns 1 and bench.py refuses to measure it
    FLAGS += EXPERIMENT + ["-DGDM_EXPERIMENT_BUILD=1"]
STAMP = os.path.join(OBJ, "flags.txt")


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm): libgdm_hip.so cannot be built")


def _newer(src, dst, extra=()):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(p) > t for p in (src, *extra))


def _flags_stamp(hipcc):
    """What the objects depend on besides their sources: the compile flags and the compiler."""
    try:
        ver = subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout.strip()
    except OSError:
        ver = "?"
    return " ".join(FLAGS) + "\n" + repr(sorted(PER_FILE_FLAGS.items())) + "\n" + ver + "\n"


def build(force=False, verbose=False):
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)
    stamp = _flags_stamp(hipcc)
    try:
        same = open(STAMP).read() == stamp
    except OSError:
        same = False
    if not same:
        force = True       # objects from another flag set (e.g. -DGDM_STAMPS experiments) must not be linked in
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    headers = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(PKG, "..", "include", "*.h"))
    jobs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o")
        if force or _newer(s, o, headers):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [hipcc, *FLAGS, *PER_FILE_FLAGS.get(os.path.basename(s), []), "-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return s, r.returncode, r.stdout + r.stderr

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        for s, rc, out in ex.map(cc, jobs):
            if verbose or rc != 0:
                sys.stderr.write(f"[build] {os.path.basename(s)} rc={rc}\n{out}\n")
            if rc != 0:
                raise RuntimeError(f"hipcc failed on {s}")
    with open(STAMP, "w") as f:
        f.write(stamp)
    objs = [os.path.join(OBJ, os.path.basename(s)[:-4] + ".o") for s in srcs]
    if force or jobs or not os.path.exists(LIB):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("link of libgdm_hip.so failed")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
