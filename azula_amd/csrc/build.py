r"""Builds ``libazula_amd.so`` (gfx950 HIP kernels + C ABI) in-tree with hipcc.

    python -m azula_amd.csrc.build [--force] [--verbose] [--regen]

No torch, no cmake: one ``hipcc -shared -fPIC`` per source file (cached on mtime) and a final link.
The resulting ``.so`` sits next to the sources so that it travels with the tree.
"""

from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SOURCES = ["transition.hip", "layout.hip", "linear.hip", "norm.hip", "conv.hip", "wino_x3.hip", "attention.hip", "graph.hip", "rng.hip", "calib.hip"]
LIB = os.path.join(HERE, "libazula_amd.so")
OBJ_DIR = os.path.join(HERE, "_obj")
FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-ffp-contract=on",  # contraction only where written as a*b+c in one expression; az_mul/az_add never fuse
    "-Wall",
    "-Wno-unused-function",
]


# Per-source extra flags.  attention.hip: keep the MFMA accumulators in VGPRs (gfx950 has a unified register file): the
# online-softmax rescale of the output accumulators otherwise costs 2 x 32 v_accvgpr moves per 32-key tile, and vector
# instructions do not overlap with fp32 MFMAs.
EXTRA_FLAGS = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: azula_amd needs the ROCm toolchain to build its gfx950 kernels")
    return exe


_FLAG_OK: dict = {}


def _supported(flags: list[str]) -> bool:
    r"""Hidden LLVM options (``-mllvm ...``) abort hipcc versions that lack them ("Unknown command line argument"): probe once
    with an empty translation unit and build without the flag where it is unknown (a slower schedule, not a broken library)."""
    key = tuple(flags)
    if key not in _FLAG_OK:
        import tempfile

        with tempfile.TemporaryDirectory(prefix="azula_amd_probe_") as tmp:  # (never inside the source tree)
            probe = os.path.join(tmp, "flag_probe.hip")
            with open(probe, "w") as f:
                f.write("#include <hip/hip_runtime.h>\n__global__ void az_flag_probe() {}\n")
            res = subprocess.run([hipcc(), "--offload-arch=gfx950", *flags, "-x", "hip", "-c", probe, "-o", probe + ".o"],
                                 capture_output=True)
        _FLAG_OK[key] = res.returncode == 0
    return _FLAG_OK[key]


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, regen: bool = False) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    headers = [os.path.join(HERE, "common.h"), os.path.join(HERE, "conv_shared.h"), os.path.join(ROOT, "include", "azula_amd.h"), os.path.join(HERE, "wino_kloop.inc")]
    headers.append(os.path.join(HERE, "igemm_kloop.inc"))
    # The hand-scheduled K loops are COMMITTED sources.  A build never rewrites them (mtime order after a checkout is arbitrary
    # and an install may be read-only); it only writes one that is missing.  Developers regenerate with `--regen` (or by running
    # the generator), and tests/test_cabi.py checks that the committed text is exactly what the generators emit.
    for inc, gen in ((headers[-2], "gen_wino_kloop.py"), (headers[-1], "gen_igemm_kloop.py")):
        if regen or not os.path.exists(inc):
            env = {k: v for k, v in os.environ.items() if k != "AZ_KLOOP_AB"}
            subprocess.run([sys.executable, os.path.join(HERE, gen)], check=True, stdout=subprocess.DEVNULL, env=env)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
    objs, jobs = [], []
    for s in srcs:  # (the translation units compile side by side: conv.hip and wino_x3.hip take a minute each)
        src = os.path.join(HERE, s)
        obj = os.path.join(OBJ_DIR, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src, *headers]):
            extra = EXTRA_FLAGS.get(s, [])
            if extra and not _supported(extra):
                print(f"[azula_amd.build] {s}: this hipcc rejects {' '.join(extra)}; building without it", file=sys.stderr)
                extra = []
            cmd = [hipcc(), *FLAGS, *extra, "-x", "hip", "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            jobs.append((cmd, subprocess.Popen(cmd)))
    for cmd, proc in jobs:
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
    if force or _stale(LIB, objs):
        cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or True, regen="--regen" in sys.argv))
