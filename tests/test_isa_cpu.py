"""CPU: what the kernels' speed depends on, checked in the gfx950 assembly hipcc emits (no GPU needed; ~25 s).

Results do not depend on any of this - the parity tests cover those - but each item below was a silent slowdown at some point of
round 5, found only by reading the disassembly, and a toolchain bump can bring it back:

  * the all-pairs kernel (csrc/gemm_hl7.h) keeps 240 of 256 registers in accumulators and fragments: a spill INSIDE the k loop
    (scratch_load / scratch_store between the MFMAs, with an s_waitcnt vmcnt(0) in front of every reload) ruins it.  hipcc did
    exactly that until the DMA row offsets stopped being loop-invariant;
  * the returning atomic that pulls the next tile's ticket must stay ONE global_atomic_add ... sc0 (hipcc's atomic optimiser
    turns a uniform-address add into a wave reduction + v_readfirstlane), and the instruction behind it must not be the wait
    for it (hipcc hoisted the id arithmetic up to the atomic once);
  * the hi / lo split of hl32 is v_fma_mix{lo,hi}_f16 (1.5 instructions per element), not v_cvt_f32_f16 + subtract + convert.
"""
import os
import re
import subprocess

import pytest

from conftest import ROOT

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CSRC = os.path.join(ROOT, "anomaly-clustering_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-Wno-unused-function",
         "-I", os.path.join(ROOT, "include")]            # models/acnative/build.py's flags


def _asm(tmp_path, unit):
    """gfx950 assembly of one translation unit, split into {kernel name: [instruction lines]}."""
    r = subprocess.run([HIPCC] + FLAGS + ["-S", "--cuda-device-only", os.path.join(CSRC, unit + ".hip"), "-o", str(tmp_path / (unit + ".s"))],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    kernels, cur = {}, None
    for line in open(tmp_path / (unit + ".s")):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = kernels.setdefault(m.group(1), [])
        elif cur is not None:
            t = line.strip()
            if t.startswith(".end_amdhsa_kernel") or t == "s_endpgm":
                cur = None if t.startswith(".end") else cur
            elif t and not t.startswith((";", ".")) or re.match(r"^\.LBB\d+_\d+:", t):
                cur.append(t)
    return kernels


def _blocks(lines):
    out, cur = [], []
    for t in lines:
        if re.match(r"^\.LBB\d+_\d+:", t):
            out.append(cur)
            cur = []
        else:
            cur.append(t)
    out.append(cur)
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_all_pairs_kernels_keep_their_k_loops_in_registers_and_their_ticket_in_flight(tmp_path):
    k = _asm(tmp_path, "alpha")
    g7 = [v for n, v in k.items() if n.startswith("_ZN5gemm76kernelI10EpiPairMin")]
    g2 = [v for n, v in k.items() if n.startswith("_ZN5gemm26kernelI10EpiPairMin")]
    assert len(g7) == 1 and len(g2) == 1
    for name, lines, mfma_per_stage in (("gemm7", g7[0], 72), ("gemm2", g2[0], 48)):
        hot = [b for b in _blocks(lines) if sum(t.startswith("v_mfma") for t in b) >= mfma_per_stage // 2]
        assert hot, name
        for b in hot:                                              # the k loop's blocks: MFMAs, LDS reads, DMA - and no scratch traffic
            assert not any(t.startswith("scratch_") for t in b), "%s: spill inside the k loop" % name
        tickets = [i for i, t in enumerate(lines) if t.startswith("global_atomic_add") and "sc0" in t]
        assert len(tickets) >= 2, "%s: the ticket atomic was rewritten" % name      # first tile + tile loop (+ steals)
    # round 6: the 256x192 kernel's transfers are buffer loads with SCALAR burst / stage offsets - no 64-bit address arithmetic and no
    # waterfall loop (a descriptor hipcc cannot prove uniform: v_readfirstlane x 4 + s_and_saveexec round every transfer) in the k loop,
    # at most 24 vector instructions beside the MFMAs and the accumulator flush in a stage block (round 5: 83)
    hot7 = [b for b in _blocks(g7[0]) if sum(t.startswith("v_mfma") for t in b) >= 36]
    assert sum(sum(bool(re.match(r"buffer_load_dwordx4 .* lds$", t)) for t in b) for b in _blocks(g7[0])) >= 14 * 2, "gemm7: the DMA is not buffer_load ... lds"
    dma_blocks = [b for b in hot7 if sum(" lds" in t and t.startswith("buffer_load") for t in b) >= 14]     # (a tile's prologue issues 36 too: not the k loop)
    assert dma_blocks, "gemm7: no k-loop block issues a whole stage's transfers"
    for b in hot7:
        assert not any(t.startswith(("v_mad_u64_u32", "v_lshl_add_u64", "s_and_saveexec")) for t in b), "gemm7: per-transfer address arithmetic / waterfall in the k loop"
    for b in dma_blocks:
        valu = [t for t in b if t.startswith("v_") and not t.startswith(("v_mfma", "v_pk_add_f32"))]
        assert len(valu) <= 24, "gemm7: %d vector instructions in the DMA block of a stage" % len(valu)
    # the tile loop's ticket of the 256x192 kernel travels under the last stages: nothing waits for it on the spot
    lines = g7[0]
    in_loop = [i for i, t in enumerate(lines) if t.startswith("global_atomic_add") and "sc0" in t]
    overlapped = [i for i in in_loop if not any(t.startswith("s_waitcnt vmcnt(0)") for t in lines[i + 1:i + 4])]
    assert overlapped, "every returning ticket atomic of gemm7 is waited for at once"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_the_hl32_split_is_fma_mix(tmp_path):
    k = _asm(tmp_path, "vit")
    att = [v for n, v in k.items() if "attention_kernel" in n]
    assert len(att) == 1
    ops = [t.split()[0] for t in att[0]]
    assert ops.count("v_fma_mixlo_f16") >= 16 and ops.count("v_fma_mixhi_f16") >= 16
    assert not any(o.startswith("v_cvt_f32_f16") for o in ops), "attention converts fp16 back to fp32: the plain split is back"
    ln = [v for n, v in k.items() if "layernorm_kernel" in n]          # (the instantiation that writes hl32; the other has no split)
    assert ln and any(any(t.startswith("v_fma_mixlo_f16") for t in v) for v in ln)
