"""CPU: the C-ABI library builds, loads, and exports every symbol include/ac_hip.h declares
(no compute calls - there is no GPU here), and argument validation fails loudly."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "ac_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ac_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import acnative
    lib = acnative.lib()                       # binds argtypes for every entry of SIGNATURES
    names = _declared()
    assert len(names) >= 18
    for name in names:
        assert hasattr(lib, name), "libac_hip.so does not export %s" % name
    assert sorted(acnative.SIGNATURES) == names, "binding table and header disagree"
    assert lib.ac_version() == acnative.ABI_VERSION == 600      # include/ac_hip.h AC_ABI_VERSION


def test_header_cites_reference_for_every_compute_entry_point():
    text = open(os.path.join(ROOT, "include", "ac_hip.h")).read()
    for ref in ("vision_transformer.py:196-207", "vision_transformer.py:107-113", "patchcore.py:375-427",
                "common.py:145-170", "utils.py:222-237", "utils.py:246-254", "main.py:294-296", "test.py:193-195"):
        assert ref in text, ref


def test_argument_validation_without_gpu():
    """Shape checks run before any HIP call, so they are testable on the CPU box."""
    import acnative
    lib = acnative.lib()
    rc = lib.ac_pack_hl32(ctypes.c_void_p(16), 4, 48, 48, ctypes.c_void_p(16), None, None)
    assert rc == 1 and b"multiple of 32" in lib.ac_last_error()
    rc = lib.ac_pair_min_dist(ctypes.c_void_p(16), 2, ctypes.c_void_p(16), 2, 8, 40, 0, 0, ctypes.c_void_p(16),
                              ctypes.c_void_p(16), 1 << 20, None)
    assert rc == 1 and b"D=40" in lib.ac_last_error()
    rc = lib.ac_pair_min_dist(ctypes.c_void_p(16), 2, ctypes.c_void_p(16), 2, 8, 64, 0, 0, ctypes.c_void_p(16),
                              ctypes.c_void_p(16), 16, None)
    assert rc == 2 and b"workspace" in lib.ac_last_error()
    assert lib.ac_alpha_softmax(ctypes.c_void_p(16), 2, 8, -1.0, 1.0, ctypes.c_void_p(16), None) == 1
    assert lib.ac_alpha_softmax_f32(ctypes.c_void_p(16), 2, 8, -1.0, 1.0, ctypes.c_void_p(16), None, None) == 1
    assert lib.ac_pair_min_dist_workspace_bytes(167, 167, 784, 4096) > 2 * 167 * 784 * 4096 * 4 // 2
    assert lib.ac_vit_workspace_bytes(32, 785, 768, 3072, 12, 192) > 32 * 785 * 3072 * 4


def test_product_refuses_cpu_tensors():
    import torch
    import acnative
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        acnative.pdist(torch.zeros(4, 8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        acnative.patch_aggregate([torch.zeros(1, 5, 8)], 3, 8, 8)
    # round-2 entry points: input pipeline, staged alpha blocks, clustering
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        acnative.preprocess_images(torch.zeros(1, 300, 300, 3, dtype=torch.uint8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        acnative.pack_images(torch.zeros(2, 8, 64))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        acnative.ward_linkage(torch.zeros(3, dtype=torch.float64), 3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        acnative.cluster_scores(torch.zeros(4, dtype=torch.int32), torch.zeros(4, dtype=torch.int32), 1, 1)
    from patchcore.clustering import cluster_metrics
    with pytest.raises(RuntimeError, match="no CPU fallback"):                  # the whole clustering chain, host data
        cluster_metrics(torch.zeros(6, 8).numpy(), ["a", "b", "a", "b", "a", "b"], device=torch.device("cpu"))


def test_argument_validation_of_the_round2_entry_points():
    """Shape / range checks that must fail BEFORE anything is launched (no GPU here)."""
    import acnative
    lib = acnative.lib()
    m3 = (ctypes.c_float * 3)(0.5, 0.5, 0.5)
    assert lib.ac_preprocess_images(ctypes.c_void_p(16), 1, 50, 400, 100, 224, m3, m3, ctypes.c_void_p(16), ctypes.c_void_p(16),
                                    1 << 20, None) == 1 and b"smaller than the crop" in lib.ac_last_error()
    assert lib.ac_preprocess_workspace_bytes(4, 900, 900, 256, 224) > 4 * 224 * 224 * 3
    assert lib.ac_ward_linkage(ctypes.c_void_p(16), 50000, ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16), 1 << 30,
                               None) == 1 and b"out of range" in lib.ac_last_error()
    assert lib.ac_ward_linkage(ctypes.c_void_p(16), 100, ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16), 64,
                               None) == 2
    assert lib.ac_ward_workspace_bytes(10000) >= 8 * 10000 * 10000
    assert lib.ac_pair_accumulate(ctypes.c_void_p(16), 4, 2, 8, 4, 0, 0, None, ctypes.c_void_p(16), None) == 1   # ld < ncols
    assert lib.ac_pair_accumulate(ctypes.c_void_p(16), 4, 8, 8, 4, 0, 0, None, ctypes.c_void_p(16), None) == 1   # mean needs shift
    assert lib.ac_pair_min_sq_packed(ctypes.c_void_p(16), ctypes.c_void_p(16), 2, ctypes.c_void_p(16), ctypes.c_void_p(16), 2, 8, 64,
                                     ctypes.c_void_p(16), ctypes.c_void_p(32), 0, 1, None) == 1                   # symmetric + minbuf_t
    assert b"symmetric case" in lib.ac_last_error()
    assert lib.ac_set_gemm_policy(77) == 1 and lib.ac_set_gemm_policy(0) == 0


def test_product_never_imports_the_oracle():
    """Nothing under anomaly-clustering_amd/ may reference the oracle package."""
    bad = []
    for root, _, files in os.walk(os.path.join(ROOT, "anomaly-clustering_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(root, f), errors="ignore").read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M):
                    bad.append(os.path.join(root, f))
    assert not bad, bad
