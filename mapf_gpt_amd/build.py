"""Build libmapf_gpt_amd.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m mapf_gpt_amd.build [--force] [--verbose]

One object per .hip translation unit (parallel), then one shared library next to the sources.
The built .so is git-ignored but travels to the GPU box with the snapshot.
"""
import concurrent.futures
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(CSRC, "_build")
LIB = os.path.join(CSRC, "libmapf_gpt_amd.so")
SOURCES = ["prof.hip", "tokenizer.hip", "env.hip", "gpt.hip", "gpt_fast.hip", "step.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-Wno-unused-but-set-variable"]


def _sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _stamp():
    h = hashlib.sha256()
    for root, _, files in os.walk(CSRC):
        if "_build" in root:
            continue
        for f in sorted(files):
            if f.endswith((".hip", ".h", ".cpp")):
                with open(os.path.join(root, f), "rb") as fh:
                    h.update(f.encode() + fh.read())
    with open(os.path.join(os.path.dirname(HERE), "include", "mapf_gpt_amd.h"), "rb") as fh:
        h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _compile(src, verbose):
    obj = os.path.join(BUILD, src.replace(".hip", ".o"))
    cmd = [HIPCC] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr)
    return obj


def build(force=False, verbose=False):
    os.makedirs(BUILD, exist_ok=True)
    stamp_file = os.path.join(BUILD, "stamp")
    stamp = _stamp()
    if not force and os.path.exists(LIB) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return LIB
    srcs = _sources()
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, verbose), srcs))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(stamp_file, "w") as f:
        f.write(stamp)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv))
