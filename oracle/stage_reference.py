"""TEST INFRASTRUCTURE ONLY - recipe that makes the *reference itself* available on the GPU box.

The reference is pure Python: there is nothing to compile into `oracle/_ref/`.  What the GPU-side checks need is the reference's
own `llava` package (its `MobileCLIPVisionTower`, `LlavaQwen2ForCausalLM`, `prepare_inputs_labels_for_multimodal`) executing on
PyTorch-ROCm next to our tower.  `/root/reference` exists only in the build container, so this recipe - run by
`__graft_entry__.build()` whenever `/root/reference` is present - packs `llava/**/*.py` and the model-config JSON files, unmodified,
into ONE archive `oracle/_ref/reference_llava.zip`.  `oracle/_ref/` is git-ignored (no reference source ever enters the history) but
not gpurun-ignored, so the archive travels to the GPU box like the built `libfvhd.so` does.  At test time `unpack()` extracts it
into a scratch directory OUTSIDE the repository and `oracle/ref_import.py` imports from there.

Only tests/, `__graft_entry__.smoke()` and bench.py's `cpu_baseline` leg may use this (the product never imports `oracle/`).

    python -m oracle.stage_reference          # (re)build the archive from /root/reference
"""
from __future__ import annotations

import hashlib
import os
import tempfile
import zipfile

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCE_ROOT = "/root/reference"
ARCHIVE = os.path.join(HERE, "_ref", "reference_llava.zip")
_KEEP = (".py", ".json")


def stage(source_root: str = SOURCE_ROOT, archive: str = ARCHIVE) -> str | None:
    """Pack <source_root>/llava (sources + configs, byte for byte) into `archive`; None when the reference is not mounted."""
    pkg = os.path.join(source_root, "llava")
    if not os.path.isdir(pkg):
        return None
    files = []
    for d, dirs, names in os.walk(pkg):
        dirs[:] = sorted(x for x in dirs if x != "__pycache__")
        for n in sorted(names):
            if n.endswith(_KEEP):
                files.append(os.path.join(d, n))
    os.makedirs(os.path.dirname(archive), exist_ok=True)
    tmp = archive + ".tmp"
    with zipfile.ZipFile(tmp, "w", zipfile.ZIP_DEFLATED) as z:
        for f in files:
            info = zipfile.ZipInfo(os.path.relpath(f, source_root), date_time=(1980, 1, 1, 0, 0, 0))   # reproducible archive
            info.compress_type = zipfile.ZIP_DEFLATED
            with open(f, "rb") as fh:
                z.writestr(info, fh.read())
    os.replace(tmp, archive)
    return archive


def unpack(archive: str = ARCHIVE) -> str | None:
    """Extract the staged archive once per content hash into the system scratch directory; returns the directory that holds
    `llava/` (to be put on sys.path), or None when nothing was staged."""
    if not os.path.isfile(archive):
        return None
    with open(archive, "rb") as fh:
        tag = hashlib.sha256(fh.read()).hexdigest()[:16]
    root = os.path.join(tempfile.gettempdir(), f"fvhd_reference_{tag}")
    marker = os.path.join(root, ".complete")
    if not os.path.isfile(marker):
        os.makedirs(root, exist_ok=True)
        with zipfile.ZipFile(archive) as z:
            z.extractall(root)
        open(marker, "w").close()
    return root


if __name__ == "__main__":
    print(stage())
