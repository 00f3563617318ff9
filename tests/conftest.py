import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`gpu` tests are skipped (not failed) on hosts without a HIP device or without the built kernel library."""
    import torch

    lib = os.path.join(ROOT, "skillful_nowcasting_amd", "lib", "libdgmr_hip.so")
    if torch.cuda.is_available() and os.path.exists(lib):
        return
    why = "no HIP device" if not torch.cuda.is_available() else "libdgmr_hip.so not built"
    skip = pytest.mark.skip(reason=f"gpu test: {why}")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    from safetensors import safe_open

    path = os.path.join(GOLDEN, name + ".safetensors")
    out = {}
    with safe_open(path, framework="pt") as f:
        for k in f.keys():
            out[k] = f.get_tensor(k)
        meta = f.metadata()
    return out, meta


def split_golden(rec):
    """-> (sd0, buf1, ins, outs, cots, grad_in, grad_p)"""
    sd0 = {k[4:]: v.clone() for k, v in rec.items() if k.startswith("sd0.")}
    buf1 = {k[5:]: v for k, v in rec.items() if k.startswith("buf1.")}
    n_in = len([k for k in rec if k.startswith("in.")])
    ins = [rec[f"in.{i}"] for i in range(n_in)]
    n_out = len([k for k in rec if k.startswith("out.")])
    outs = [rec[f"out.{i}"] for i in range(n_out)]
    cots = [rec[f"cot.{i}"] for i in range(n_out)]
    grad_in = {int(k.split(".")[-1]): v for k, v in rec.items() if k.startswith("grad.in.")}
    grad_p = {k[7:]: v for k, v in rec.items() if k.startswith("grad.p.")}
    return sd0, buf1, ins, outs, cots, grad_in, grad_p
