"""CPU tests of the C ABI: the library loads, exports every symbol include/gklt.h declares, and
its host-only arithmetic (context, kernels, borders) equals the oracle bit for bit.  No compute
call is made here (there is no GPU in this container)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "gklt.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gklt_[a-z0-9_]+)\s*\(", src)))


def test_exports_every_declared_symbol(klt):
    names = _declared()
    assert len(names) >= 30
    L = klt.lib()
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_struct_layout_matches_header(klt):
    src = open(os.path.join(ROOT, "include", "gklt.h")).read()
    body = re.search(r"typedef struct gklt_params \{(.*?)\} gklt_params;", src, re.S).group(1)
    fields = []
    for decl in re.findall(r"(int|float)\s+([^;]+);", body):
        fields += [(decl[0], n.strip()) for n in decl[1].split(",")]
    assert [n for _, n in fields] == [n for n, _ in klt.Params._fields_]
    for (t, n), (_, ct) in zip(fields, klt.Params._fields_):
        assert ct is (C.c_int if t == "int" else C.c_float), n


def test_host_arithmetic_equals_oracle(klt, ok):
    for sigma in (0.3, 0.5, 0.7, 0.9, 1.0, 1.1, 1.3, 1.5, 1.8, 2.1, 3.3, 7.0):
        g, d = klt.computeKernels(sigma)
        rg, rd = ok.compute_kernels(sigma)
        np.testing.assert_array_equal(g, rg)
        np.testing.assert_array_equal(d, rd)
    with pytest.raises(klt.GkltError):
        klt.computeKernels(40.0)
    with pytest.raises(klt.GkltError):
        klt.computeKernels(-1.0)
    a, b = klt.KLTCreateTrackingContext(), ok.default_params()
    for name, _ in klt.Params._fields_:
        assert getattr(a, name) == getattr(b, name), name
    for win in (3, 4, 5, 7, 9, 11, 15, 21, 31):
        for levels in (1, 2, 3, 4, 5):
            for ss in (2, 4):
                assert klt.make_params(window=win, levels=levels, subsampling=ss).borderx == \
                    ok.make_params(window=win, levels=levels, subsampling=ss).borderx
        for sr in (3, 8, 15, 30, 60, 200):
            p, q = klt.KLTCreateTrackingContext(), ok.default_params()
            p.window_width = p.window_height = win
            q.window_width = q.window_height = win
            assert klt.KLTChangeTCPyramid(p, sr) == ok.change_tc_pyramid(q, sr)
            assert (p.nPyramidLevels, p.subsampling, p.window_width) == (q.nPyramidLevels, q.subsampling, q.window_width)
    assert klt._KLTComputeSmoothSigma(a) == np.float32(0.1) * 21
    assert klt.KLTCountRemainingFeatures([0, -1, 5, -4, 0]) == 3


def test_no_gpu_is_a_loud_error(klt):
    """The product path has no CPU fallback: without a gfx950 device context creation fails."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(klt.GkltError):
        klt.Context(0)


def test_oracle_is_not_reachable_from_the_product():
    """Nothing under gslam_amd/ or include/ may import, link or load oracle/."""
    for base in ("gslam_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".h", ".hip", ".cpp")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    assert "klt_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f


def test_summation_order_choices_are_defined_once():
    """The 2x2 mean of the exact-half resize is an order CHOICE (OpenCV is not in the reference
    tree): kernels and oracle take it from include/gklt_order.h, nobody spells it out locally, so a
    fixture that settles the order changes one line for both sides."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    users = 0
    for f in glob.glob(os.path.join(root, "gslam_amd", "csrc", "*")) + [os.path.join(root, "oracle", "klt_oracle.c")]:
        text = open(f).read()
        code = re.sub(r"//[^\n]*|/\*.*?\*/", "", text, flags=re.S)
        assert "0.25f" not in code, "%s averages a 2x2 block without GKLT_MEAN2X2" % f
        users += code.count("GKLT_MEAN2X2(")
    assert users >= 5
    hdr = open(os.path.join(root, "include", "gklt_order.h")).read()
    assert "GKLT_MEAN2X2_PAIRWISE" in hdr and hdr.count("#define GKLT_MEAN2X2(") == 2
