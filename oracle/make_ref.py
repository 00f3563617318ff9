"""Stage the UNMODIFIED reference package for the CPU baseline:  python oracle/make_ref.py

`/root/reference` exists only in the build container; bench.py's `cpu_baseline` leg runs on the GPU box's host.  This recipe copies
the reference's own package directory (`/root/reference/dgmr`, pure Python, ~2 kLoC) byte for byte into the git-ignored
`oracle/_ref/dgmr` so that it travels with the push like the built `.so` does (`oracle/_ref/` is in .gitignore and NOT in
.gpurunignore).  Nothing is edited; `oracle/_ref/MANIFEST.json` records the sha256 of every file copied.  bench.py imports it
through `oracle/_stubs.py` (stand-ins for the three packages the image lacks) and reports `cpu_baseline.kind = "reference"`; when
`oracle/_ref` is absent it falls back to the oracle (`kind = "port"`).

Test / measurement infrastructure only: the product (`skillful_nowcasting_amd/`) never imports anything under `oracle/`
(tests/test_abi.py greps for it).  Called by `__graft_entry__.build()` whenever the source directory exists.
"""
import hashlib
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.environ.get("DGMR_REFERENCE_ROOT", "/root/reference")
DST = os.path.join(HERE, "_ref")


def stage(src_root: str = SRC, dst: str = DST) -> bool:
    """Copy <src_root>/dgmr -> <dst>/dgmr when the source exists; True when oracle/_ref/dgmr is in place afterwards."""
    pkg = os.path.join(src_root, "dgmr")
    out = os.path.join(dst, "dgmr")
    if not os.path.isdir(pkg):
        return os.path.isdir(out)
    manifest = {}
    for root, _dirs, files in os.walk(pkg):
        for f in sorted(files):
            if not f.endswith(".py"):
                continue
            p = os.path.join(root, f)
            manifest[os.path.relpath(p, src_root)] = hashlib.sha256(open(p, "rb").read()).hexdigest()
    man_path = os.path.join(dst, "MANIFEST.json")
    if os.path.isdir(out) and os.path.exists(man_path):
        try:
            if json.load(open(man_path)).get("files") == manifest:
                return True
        except (OSError, ValueError):
            pass
    if os.path.isdir(out):
        shutil.rmtree(out)
    os.makedirs(dst, exist_ok=True)
    shutil.copytree(pkg, out, ignore=shutil.ignore_patterns("__pycache__", "*.pyc"))
    with open(man_path, "w") as f:
        json.dump({"source": pkg, "note": "byte-for-byte copy of the reference package, staged by oracle/make_ref.py for "
                                          "bench.py's cpu_baseline leg; git-ignored, never imported by the product", "files": manifest}, f, indent=1)
    return True


if __name__ == "__main__":
    ok = stage()
    print("oracle/_ref/dgmr", "ready" if ok else "absent (no reference source here)")
    sys.exit(0 if ok else 1)
