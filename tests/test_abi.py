"""The C-ABI boundary: include/dgmr_hip.h <-> libdgmr_hip.so <-> the ctypes table in skillful_nowcasting_amd/_lib.py.

CPU only: the library is dlopen'ed and its symbols resolved; no kernel is launched (there is no GPU here).
"""
import ctypes
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "dgmr_hip.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)  # drop comments
    src = re.sub(r"typedef struct \w+ \{.*?\} \w+;", "", src, flags=re.S)
    names = re.findall(r"\b(dgmr_\w+)\s*\(", src)
    return sorted(set(names))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge

    ge.build()  # hipcc cross-compiles gfx950 without a GPU
    from skillful_nowcasting_amd import _lib

    return _lib.load()


def test_header_declares_something():
    names = _declared_functions()
    assert len(names) >= 40, names
    assert "dgmr_conv_fwd" in names and "dgmr_adam" in names


def test_library_exports_every_declared_symbol(lib):
    for name in _declared_functions():
        assert hasattr(lib, name), f"{name} is declared in include/dgmr_hip.h but not exported by libdgmr_hip.so"


def test_ctypes_table_matches_header(lib):
    from skillful_nowcasting_amd import _lib

    declared = set(_declared_functions())
    bound = set(_lib.SIGNATURES) | set(_lib.SIGNATURES_I64) | {"dgmr_abi_version", "dgmr_last_error", "dgmr_profile_variant_name"}
    assert declared == bound, f"header-only: {sorted(declared - bound)}  binding-only: {sorted(bound - declared)}"
    assert lib.dgmr_abi_version() == _lib.ABI_VERSION


def test_struct_layouts_match_header():
    """ConvArgs / WgradArgs mirror the C structs field for field (names and order)."""
    from skillful_nowcasting_amd import _lib

    src = open(HEADER).read()
    for cname, pyt in (("dgmr_conv_args", _lib.ConvArgs), ("dgmr_wgrad_args", _lib.WgradArgs),
                       ("dgmr_sn_desc", _lib.SNDesc), ("dgmr_adam_desc", _lib.AdamDesc)):
        m = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), src, flags=re.S)
        if m is None and pyt is None:
            continue
        assert m is not None and pyt is not None, cname
        body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            names = decl.split(None, 1)[1] if not decl.startswith("const") else decl.split(None, 2)[2]
            for nm in names.split(","):
                fields.append(nm.strip().lstrip("*").strip())
        assert fields == [f[0] for f in pyt._fields_], (cname, fields, [f[0] for f in pyt._fields_])
    # the numpy record the optimiser fills on the host is the same 56 bytes, field for field
    import ctypes

    assert _lib.ADAM_DESC_DTYPE.itemsize == ctypes.sizeof(_lib.AdamDesc)
    for name, _t in _lib.AdamDesc._fields_:
        assert _lib.ADAM_DESC_DTYPE.fields[name][1] == getattr(_lib.AdamDesc, name).offset, name


def test_argument_errors_are_reported_without_a_gpu(lib):
    """Argument validation happens before any launch: a bad call returns <0 and sets dgmr_last_error()."""
    from skillful_nowcasting_amd._lib import ConvArgs

    a = ConvArgs()  # all-null pointers
    rc = lib.dgmr_conv_fwd(ctypes.byref(a), None)
    assert rc < 0
    assert b"null pointer" in lib.dgmr_last_error()


def test_wgrad_plan_is_host_arithmetic(lib):
    """dgmr_conv_wgrad_plan fills nsplit from the geometry alone (no launch): a multiple of the call groups, at least one slab per
    group, and - in the bf16 modes, for 3x3 convs on maps made of whole rows of 32 / 16 pixels - sized for the window kernel."""
    from skillful_nowcasting_amd._lib import WgradArgs

    def plan(prec, n, h, w, cin, cout, k, groups):
        assert lib.dgmr_set_precision(prec) == 0
        a = WgradArgs()
        a.N, a.D, a.H, a.W, a.Cin, a.Cout = n, 1, h, w, cin, cout
        a.KD, a.KH, a.KW = 1, k, k
        a.groups, a.pre_group = groups, 1
        assert lib.dgmr_conv_wgrad_plan(ctypes.byref(a)) == 0
        return a.nsplit

    try:
        for prec in (0, 1, 2):
            for (n, h, w, cin, cout, k, groups) in [(288, 128, 128, 96, 96, 3, 18), (288, 16, 16, 768, 768, 3, 18), (16, 8, 8, 768, 768, 3, 1),
                                                    (288, 64, 64, 48, 96, 1, 18), (4, 32, 32, 8, 8, 3, 2), (36, 64, 64, 192, 192, 3, 18)]:
                ns = plan(prec, n, h, w, cin, cout, k, groups)
                assert ns >= groups and ns % groups == 0 and ns <= 4096, (prec, n, h, w, cin, cout, k, groups, ns)
        # window path (bf16x3, 3x3, W % 32 == 0), workgroups = chunks x output tiles x slabs: one round of <= 256 for the wave-specialised
        # kernels (one workgroup per CU), of <= 512 for the one-role kernel (dgmr_conv_tune wgrad_window = 1)
        ns = plan(1, 288, 128, 128, 96, 96, 3, 18)
        assert 3 * 1 * ns <= 256 and ns == 72
        assert lib.dgmr_conv_tune(-1, -1, -1, 1) == 0
        ns = plan(1, 288, 128, 128, 96, 96, 3, 18)
        assert lib.dgmr_conv_tune(-1, -1, -1, -1) == 0
        assert 3 * 1 * ns <= 512 and ns == 162
        # f32 mode keeps the im2col kernel's slab count
        assert plan(0, 288, 128, 128, 96, 96, 3, 18) == lib.dgmr_conv_wgrad_nsplit(288 * 128 * 128, 96, 9 * 96, 18)
    finally:
        lib.dgmr_set_precision(0)


def test_tune_and_small_op_argument_checks(lib):
    assert lib.dgmr_conv_tune(-1, -1, -1, -1) == 0
    assert lib.dgmr_conv_tune(99, -1, -1, -1) < 0 and b"dgmr_conv_tune" in lib.dgmr_last_error()
    assert lib.dgmr_conv_tune(-1, -1, -1, -1) == 0
    buf = (ctypes.c_float * 8)()
    assert lib.dgmr_repeat_rows(buf, buf, 6, 2, None) < 0 and b"multiple of 4" in lib.dgmr_last_error()
    assert lib.dgmr_group_rowsum(buf, None, buf, 1, 2, 6, 1, 1, None) < 0 and b"multiple of 4" in lib.dgmr_last_error()
    assert lib.dgmr_repeat_interleave(buf, buf, 2, 6, 2, None) < 0 and b"multiple of 4" in lib.dgmr_last_error()


def test_wgrad_plan_with_batched_draw_groups(lib):
    """6 generator draws x 18 forecast steps = 108 call groups per launch (the batched generator pass): at least one slab per
    group, for the window kernel and for the im2col kernels."""
    from skillful_nowcasting_amd._lib import WgradArgs

    try:
        for prec in (0, 1):
            assert lib.dgmr_set_precision(prec) == 0
            for (n, h, w, cin, cout, k) in [(1728, 128, 128, 96, 96, 3), (1728, 8, 8, 768, 768, 3), (1728, 64, 64, 96, 192, 1),
                                            (108, 8, 8, 768, 384, 3)]:
                a = WgradArgs()
                a.N, a.D, a.H, a.W, a.Cin, a.Cout = n, 1, h, w, cin, cout
                a.KD, a.KH, a.KW = 1, k, k
                a.groups, a.pre_group = 108, 1
                assert lib.dgmr_conv_wgrad_plan(ctypes.byref(a)) == 0
                assert a.nsplit >= 108 and a.nsplit % 108 == 0 and a.nsplit <= 4096, (prec, n, h, w, cin, cout, k, a.nsplit)
    finally:
        lib.dgmr_set_precision(0)


def test_lds_dma_stages_are_drained_before_their_barrier(lib, tmp_path):
    """conv_win_glds.h publishes every weight stage written by global_load_lds (an LDS write tracked only by the issuing wave's
    vmcnt) with `s_waitcnt vmcnt(0)` + barrier.  The wait is explicit in the source (dma_drain); this checks the BINARY: walking
    back from every s_barrier of every conv3x3_glds_kernel instantiation, an s_waitcnt with vmcnt(0) must come before any
    global_load_lds is met (a compiler or flag change that dropped or moved the wait would race silently)."""
    import shutil
    import subprocess

    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    from skillful_nowcasting_amd import _lib

    so = tmp_path / "libdgmr_hip.so"
    shutil.copy(_lib.LIB_PATH, so)
    subprocess.run([objdump, "--offloading", str(so)], check=True, cwd=tmp_path, capture_output=True)
    bundles = [f for f in os.listdir(tmp_path) if "gfx950" in f]
    assert bundles, os.listdir(tmp_path)
    checked = 0
    for bname in bundles:
        asm = subprocess.run([objdump, "-d", str(tmp_path / bname)], check=True, capture_output=True, text=True).stdout.split("\n")
        heads = [i for i, l in enumerate(asm) if l.endswith(">:")]
        for hi, start in enumerate(heads):
            if "conv3x3_glds_kernel" not in asm[start]:
                continue
            body = asm[start:heads[hi + 1] if hi + 1 < len(heads) else len(asm)]
            assert any("global_load_lds_dwordx4" in l for l in body), asm[start]
            for b in [i for i, l in enumerate(body) if "s_barrier" in l]:
                j = b - 1
                while j >= 0 and "s_barrier" not in body[j]:
                    assert "global_load_lds" not in body[j], f"{asm[start]}: LDS-DMA reaches the barrier at +{b} without vmcnt(0)"
                    if re.search(r"s_waitcnt.*vmcnt\(0\)", body[j]):
                        break
                    j -= 1
                checked += 1
    assert checked >= 30, checked  # three instantiations x (1 + 9 + 1) barriers


def test_wave_specialised_wgrad_keeps_its_prefetch_in_flight(lib, tmp_path):
    """wgrad_ws.h: the loader waves fetch tile t + 3 into one of two register sets while tile t + 1 is being stored, so at every wait
    of the loader program a whole set (XP + YP sixteen-byte loads) must be allowed to stay in flight.  hipcc computes s_waitcnt vmcnt
    as the minimum over all control-flow paths: one conditional prefetch, a break inside the loop, or a scratch reload between two
    global loads turns the waits into vmcnt(0 ... 9) and the prefetch depth of two into one - silently, with correct results and
    20 - 60 % longer launches.  Checked in the BINARY, for every instantiation: between the first and the last
    global load, no wait below XP + YP; transposing LDS reads present; no scratch instruction in the bf16 / bf16x3 variants."""
    import shutil
    import subprocess

    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    from skillful_nowcasting_amd import _lib

    so = tmp_path / "libdgmr_hip.so"
    shutil.copy(_lib.LIB_PATH, so)
    subprocess.run([objdump, "--offloading", str(so)], check=True, cwd=tmp_path, capture_output=True)
    checked = set()
    for bname in [f for f in os.listdir(tmp_path) if "gfx950" in f]:
        asm = subprocess.run([objdump, "-d", str(tmp_path / bname)], check=True, capture_output=True, text=True).stdout.split("\n")
        heads = [i for i, l in enumerate(asm) if l.endswith(">:")]
        for hi, start in enumerate(heads):
            m = re.search(r"conv_wgrad_ws_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", asm[start])
            if not m:
                continue
            bi, ns, tws, mw = (int(v) for v in m.groups())
            body = asm[start:heads[hi + 1] if hi + 1 < len(heads) else len(asm)]
            loads = [i for i, l in enumerate(body) if "global_load_dwordx4" in l]
            assert loads, asm[start]
            hpix = (32 + 2) * (2 + 2) if tws == 5 else (16 + 2) * (4 + 2)
            per_set = -(-hpix * 8 // 256) + bi // 16  # XP + YP
            waits = [int(w.group(1)) for l in body[loads[0]:loads[-1] + 1] for w in [re.search(r"s_waitcnt.*vmcnt\((\d+)\)", l)] if w]
            assert waits and min(waits) >= per_set, f"{asm[start]}: vmcnt waits {sorted(set(waits))} in the loader program, a set is {per_set} loads"
            assert any("ds_read_b64_tr_b16" in l for l in body), asm[start]
            if ns != 6:
                assert not any("scratch_" in l for l in body), f"{asm[start]}: scratch traffic in the weight-gradient kernel"
            checked.add((bi, ns, tws, mw))
    # BI {64, 96} x tile {32x2, 16x4} x (bf16, bf16x3: 3 and 4 matrix waves; bf16x6: 3) + the pixel-split 48-column tile (round 6) x tile x (bf16, bf16x3)
    assert len(checked) == 24, sorted(checked)


def test_missing_library_fails_loudly(monkeypatch):
    from skillful_nowcasting_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libdgmr_hip.so")
    with pytest.raises(RuntimeError, match="no fallback"):
        _lib.load()
