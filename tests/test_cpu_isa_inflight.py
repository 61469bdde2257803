"""The register-resident chain's inline-asm LDS reads: no instruction may touch a destination register before the wait that
covers it (scripts/isa_inflight_check.py explains the hazard; it produced wrong values during bring-up).  Compiles the kernel
for gfx950 (hipcc cross-compiles without a GPU) and scans the listing."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
sys.path.insert(0, ROOT)
from __graft_entry__ import HIPCC_FLAGS          # the listings scanned here are compiled with the SHIPPING flags

LISTING = ["hipcc"] + [f for f in HIPCC_FLAGS if f != "-fPIC"] + ["-S", "--cuda-device-only"]
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def test_scanner_sees_both_hazard_kinds():
    from isa_inflight_check import scan
    use_before_wait = ["ds_read_b64 v[6:7], v52 offset:96", "v_add_f32_e32 v32, v25, v6", "s_waitcnt lgkmcnt(0)"]
    clobber = ["ds_read_b64 v[12:13], v185", "v_mul_f32_e64 v12, |v7|, s51", "s_waitcnt lgkmcnt(0)"]
    clean = ["ds_read_b128 v[96:99], v193 offset:0x4800", "v_mfma_f32_32x32x16_bf16 a[0:15], v[1:4], v[5:8], a[0:15]",
             "s_waitcnt lgkmcnt(0)", "v_add_f32_e32 v1, v96, v97"]
    assert [f[3] for f in scan(use_before_wait)[1]] == ["read"]
    assert [f[3] for f in scan(clobber)[1]] == ["write"]
    assert scan(clean) == (1, [])


def test_scanner_understands_counted_waits():
    """lgkmcnt(N) completes all but the N youngest LDS operations of the wave (they return in order) -- the training chain leaves its
    patch writes in flight that way; a scalar memory operation in between makes the count meaningless."""
    from isa_inflight_check import scan
    head = ["ds_read_b128 v[10:13], v1", "ds_write_b32 v2, v3", "ds_write_b32 v2, v4 offset:160"]
    assert scan(head + ["s_waitcnt lgkmcnt(2)", "v_mfma_f32_32x32x16_f16 a[0:15], v[10:13], v[20:23], a[0:15]"])[1] == []
    late = scan(head + ["s_waitcnt lgkmcnt(3)", "v_mfma_f32_32x32x16_f16 a[0:15], v[10:13], v[20:23], a[0:15]"])[1]
    assert late and all(f[3] == "read" for f in late)
    # the read is the YOUNGEST operation: a count of 1 leaves it in flight
    young = scan(["ds_write_b32 v2, v3", "ds_read_b128 v[10:13], v1", "s_waitcnt lgkmcnt(1)", "v_add_f32 v5, v10, v11"])[1]
    assert young and all(f[3] == "read" for f in young)
    smem = scan(head[:1] + ["s_load_dwordx2 s[4:5], s[0:1], 0x0", "ds_write_b32 v2, v3", "s_waitcnt lgkmcnt(1)", "v_add_f32 v5, v10, v11"])[1]
    assert smem and all(f[3] == "read" for f in smem)
    assert scan(head[:1] + ["s_load_dwordx2 s[4:5], s[0:1], 0x0", "s_waitcnt lgkmcnt(0)", "v_add_f32 v5, v10, v11"])[1] == []


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not installed")
def test_chain_t_listing_is_clean(tmp_path):
    from isa_inflight_check import scan
    out = tmp_path / "chain_t.s"
    subprocess.run(LISTING + [os.path.join(ROOT, "dynhor_amd", "csrc", "chain_t.hip"), "-o", str(out)], check=True, timeout=600,
                   stderr=subprocess.DEVNULL)
    text = out.read_text()
    n_reads, found = scan(text.split("\n"))
    assert n_reads > 1000, "the listing does not look like the chain kernel"
    assert not found, found[:5]
    import re
    kernels = re.findall(r"^(_ZN2dh\w+):.*?; ScratchSize: (\d+).*?; LDSByteSize: (\d+)", text, flags=re.S | re.M)
    names = {k[0]: (int(k[1]), int(k[2])) for k in kernels}
    nograd = [v for k, v in names.items() if "sdf_nograd_t_kernel" in k]
    train = [v for k, v in names.items() if "sdf_fwd_train_t_kernel" in k]
    assert nograd and train, names
    # the no-grad kernel must not spill at all; the training forward keeps a few loop invariants in scratch, reloaded outside the
    # MFMA stream (the scanner above would flag a spilled in-flight register: scratch stores read their data register)
    assert nograd[0][0] == 0 and train[0][0] <= 64, names
    # the shipping (two-piece fp16) forms: no scratch at all
    nograd_h = [v for k, v in names.items() if "sdf_nograd_h_kernel" in k]
    train_h = [v for k, v in names.items() if "sdf_fwd_train_h_kernel" in k]
    assert nograd_h and train_h, names
    assert nograd_h[0][0] == 0 and train_h[0][0] == 0, names
    for scratch, lds in nograd + train + nograd_h + train_h:
        assert lds <= 160 * 1024, "one workgroup per CU: the LDS image must fit 160 KB"
    _no_register_soffset_on_wide_stores(text)
    # Round 6: the fp16 training forward transposes its m-tiles on the MATRIX pipe (csrc/chain_t.hip T_SAVE_MFMA).  What the form is
    # worth rests on three things the compiler does today -- guard them: the selector MFMAs are there (4 per saved m-tile on top of
    # the 2,016 of the patch form), the LDS patch is gone from the dealt stream (the patch form had 771 ds_write_b32; what remains is the
    # embedding image and the once-per-tile feature tile), and the tile stores read the transposed tile straight from the accumulator
    # registers it was formed in (no copy into vector registers)
    m = re.search(r"^_ZN2dh22sdf_fwd_train_h_kernel\w*:[^\n]*\n(.*?)s_endpgm", text, flags=re.S | re.M)
    assert m, "sdf_fwd_train_h_kernel not found in the listing"
    body = m.group(1)
    n_mfma = len(re.findall(r"^\s+v_mfma_f32_32x32x16_f16 ", body, flags=re.M))
    n_w32 = len(re.findall(r"^\s+ds_write_b32 ", body, flags=re.M))
    n_acc_stores = len(re.findall(r"^\s+buffer_store_dwordx4 a\[", body, flags=re.M))
    assert n_mfma >= 2016 + 4 * 39, n_mfma
    assert n_w32 <= 200, n_w32
    assert n_acc_stores >= 4 * 39, n_acc_stores


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not installed")
def test_tile_resident_chains_fit_two_workgroups_per_cu_and_do_not_spill_in_their_layer_loops(tmp_path):
    """The five tile-resident chains of the shipping arithmetic (kernels_mlp_h.hip): the piece-plane LDS image + aux image must
    leave room for TWO workgroups per CU (the second one's epilogue overlaps the first one's MFMAs), and whatever the compiler
    keeps in scratch must stay outside the GEMM loops (no scratch access between two MFMAs of a k-chunk)."""
    import re
    out = tmp_path / "chains_h.s"
    subprocess.run(LISTING + [os.path.join(ROOT, "dynhor_amd", "csrc", "kernels_mlp_h.hip"), "-o", str(out)], check=True, timeout=600,
                   stderr=subprocess.DEVNULL)
    text = out.read_text()
    kernels = re.findall(r"^(_ZN2dh\w+):(.*?); ScratchSize: (\d+).*?; LDSByteSize: (\d+)", text, flags=re.S | re.M)
    want = ("color_fwd_h_kernel", "sdf_grad_h_kernel", "color_bwd_h_kernel", "sdf_tangent_h_kernel", "sdf_bwd_h_kernel")
    seen = set()
    for name, body, scratch, lds in kernels:
        hit = [w for w in want if w in name]
        if not hit:
            continue
        seen.add(hit[0])
        assert 2 * int(lds) <= 160 * 1024, (name, lds)
        # inside a run of MFMAs (one k-chunk = 12 of them) nothing may touch scratch
        lines = [l.strip() for l in body.split("\n")]
        idx = [i for i, l in enumerate(lines) if l.startswith("v_mfma")]
        for a, b in zip(idx, idx[1:]):
            if b - a <= 6:
                assert not any(l.startswith("scratch_") for l in lines[a:b]), (name, lines[a:b])
    assert seen == set(want), seen
    _no_register_soffset_on_wide_stores(text)


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not installed")
def test_pair_chains_keep_scratch_out_of_their_phases_and_fit_one_workgroup_per_cu(tmp_path):
    """The tile-pair chains (chain_pair.hip, round 6): one workgroup per CU -- two piece-plane images + two aux images within 160 KB --
    and a weight slice held in registers across a layer's two phases.  A phase is 192 MFMAs with the other tile's epilogue dealt
    between them: whatever the compiler keeps in scratch (a few loop invariants; the slice itself once was: 250 registers through
    scratch and back, until the first layer's slice was requested per pair) must stay OUTSIDE the phases -- a scratch reload inside
    one waits on vmcnt(0) and drains every tile load, tile store and weight reload in flight, with ONE wave per SIMD to hide it."""
    import re
    out = tmp_path / "chain_pair.s"
    subprocess.run(LISTING + [os.path.join(ROOT, "dynhor_amd", "csrc", "chain_pair.hip"), "-o", str(out)], check=True, timeout=900,
                   stderr=subprocess.DEVNULL)
    text = out.read_text()
    kernels = re.findall(r"^(_ZN2dh\w+):(.*?); ScratchSize: (\d+).*?; LDSByteSize: (\d+)", text, flags=re.S | re.M)
    want = ("color_fwd_p_kernel", "sdf_grad_p_kernel", "color_bwd_p_kernel")
    seen = set()
    for name, body, scratch, lds in kernels:
        hit = [w for w in want if w in name]
        if not hit:
            continue
        seen.add(hit[0])
        assert int(lds) <= 160 * 1024, (name, lds)
        assert int(scratch) <= 128, (name, scratch)
        lines = [l.strip() for l in body.split("\n")]
        idx = [i for i, l in enumerate(lines) if l.startswith("v_mfma")]
        assert len(idx) >= 3 * 192, (name, len(idx))
        # runs of MFMAs no more than 40 lines apart = phases (a k-chunk boundary carries 4 LDS reads, 4 weight loads and waits)
        runs, start = [], 0
        for j in range(1, len(idx) + 1):
            if j == len(idx) or idx[j] - idx[j - 1] > 40:
                runs.append((idx[start], idx[j - 1], j - start))
                start = j
        phases = [r for r in runs if r[2] >= 150]
        # between two MFMAs of a k-chunk (its 12 slots carry a few epilogue instructions each) nothing touches scratch -- in any pair
        # kernel; the form that SHIPS by default (colour forward) keeps its phases free of scratch altogether, chunk boundaries included
        # (the other two reload one loop invariant between two phases)
        for a, b in zip(idx, idx[1:]):
            if b - a <= 12:
                assert not any(l.startswith("scratch_") for l in lines[a:b]), (name, lines[a:b])
        if "color_fwd_p_kernel" in name:
            assert len(phases) >= 2 and sum(r[2] for r in phases) >= 3 * 192, (name, [r[2] for r in runs])
            for a, b, n in phases:
                bad = [l for l in lines[a:b] if l.startswith("scratch_")]
                assert not bad, (name, n, bad[:3])
    assert seen == set(want), seen
    _no_register_soffset_on_wide_stores(text)


def _no_register_soffset_on_wide_stores(text):
    """hipcc (ROCm 7.2) guards the data registers of a buffer_store_dwordx3/x4 against an immediately following vector write only
    when soffset is an immediate; with a REGISTER soffset gfx950 stored the overwritten values (profiles/r04_ab_chain_io.json).
    Every wide buffer store of the library therefore carries its offset in voffset + immediate."""
    import re
    bad = [l.strip() for l in text.split("\n") if re.match(r"\s*buffer_store_dwordx[34] .*\], s\d+ ", l)]
    assert not bad, bad[:5]


PACKED_FP32 = ("v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32", "v_pk_mov_b32")


def test_packed_fp32_pattern_is_recognised():
    """The trap (csrc/layout.h, profiles/r05_dw_aux_hazard_table.json): packed-fp32 VALU instructions that broadcast one dword of a
    register pair through op_sel -- hipcc's code for `vector * scalar` -- are occasionally wrong in lanes 16-31 / 48-63 on gfx950."""
    failing = ["v_pk_mul_f32 v[50:51], v[38:39], v[56:57] op_sel:[0,1]",
               "v_pk_fma_f32 v[94:95], v[38:39], v[56:57], v[42:43] op_sel:[0,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]"]
    assert all(any(l.startswith(p) for p in PACKED_FP32) for l in failing)


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump not installed")
def test_built_library_holds_no_packed_fp32_instruction_and_no_function_call(tmp_path):
    """Disassembles every gfx950 code object of the library AS BUILT (what travels to the GPU box), so a build whose script forgot the
    -packed-fp32-ops flags cannot ship unnoticed.  Also: no s_swappc -- everything is inlined (a per-kernel target attribute, the
    first attempt at switching the instruction class off, silently turned the HIP headers' shuffles into 456 real calls)."""
    from dynhor_amd import _lib
    lib = tmp_path / "lib.so"
    shutil.copy(_lib.LIB_PATH, lib)
    subprocess.run([OBJDUMP, "--offloading", str(lib)], check=True, cwd=tmp_path, capture_output=True, timeout=300)
    objs = sorted(p for p in os.listdir(tmp_path) if p.endswith("gfx950"))
    assert len(objs) >= 10, objs
    n_mfma, bad, calls = 0, [], 0
    kernel = None
    for o in objs:
        dis = subprocess.run([OBJDUMP, "-d", str(tmp_path / o)], check=True, capture_output=True, text=True, timeout=300).stdout
        for line in dis.split("\n"):
            t = line.strip()
            if t.endswith(">:"):
                kernel = t
            elif t.startswith("v_mfma"):
                n_mfma += 1
            elif t.startswith(PACKED_FP32):
                bad.append((kernel, t[:90]))
            elif t.startswith("s_swappc"):
                calls += 1
    assert n_mfma > 5000, "this does not look like the library"
    assert not bad, f"{len(bad)} packed-fp32 instructions in the built library, e.g. {bad[:3]}"
    assert calls == 0, f"{calls} function calls in the built library"
