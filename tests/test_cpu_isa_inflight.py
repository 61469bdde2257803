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


def test_scanner_sees_both_hazard_kinds():
    from isa_inflight_check import scan
    use_before_wait = ["ds_read_b64 v[6:7], v52 offset:96", "v_add_f32_e32 v32, v25, v6", "s_waitcnt lgkmcnt(0)"]
    clobber = ["ds_read_b64 v[12:13], v185", "v_mul_f32_e64 v12, |v7|, s51", "s_waitcnt lgkmcnt(0)"]
    clean = ["ds_read_b128 v[96:99], v193 offset:0x4800", "v_mfma_f32_32x32x16_bf16 a[0:15], v[1:4], v[5:8], a[0:15]",
             "s_waitcnt lgkmcnt(0)", "v_add_f32_e32 v1, v96, v97"]
    assert [f[3] for f in scan(use_before_wait)[1]] == ["read"]
    assert [f[3] for f in scan(clobber)[1]] == ["write"]
    assert scan(clean) == (1, [])


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not installed")
def test_chain_t_listing_is_clean(tmp_path):
    from isa_inflight_check import scan
    out = tmp_path / "chain_t.s"
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                    os.path.join(ROOT, "dynhor_amd", "csrc", "chain_t.hip"), "-o", str(out)], check=True, timeout=600)
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


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not installed")
def test_tile_resident_chains_fit_two_workgroups_per_cu_and_do_not_spill_in_their_layer_loops(tmp_path):
    """The five tile-resident chains of the shipping arithmetic (kernels_mlp_h.hip): the piece-plane LDS image + aux image must
    leave room for TWO workgroups per CU (the second one's epilogue overlaps the first one's MFMAs), and whatever the compiler
    keeps in scratch must stay outside the GEMM loops (no scratch access between two MFMAs of a k-chunk)."""
    import re
    out = tmp_path / "chains_h.s"
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                    os.path.join(ROOT, "dynhor_amd", "csrc", "kernels_mlp_h.hip"), "-o", str(out)], check=True, timeout=600)
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


def _no_register_soffset_on_wide_stores(text):
    """hipcc (ROCm 7.2) guards the data registers of a buffer_store_dwordx3/x4 against an immediately following vector write only
    when soffset is an immediate; with a REGISTER soffset gfx950 stored the overwritten values (profiles/r04_ab_chain_io.json).
    Every wide buffer store of the library therefore carries its offset in voffset + immediate."""
    import re
    bad = [l.strip() for l in text.split("\n") if re.match(r"\s*buffer_store_dwordx[34] .*\], s\d+ ", l)]
    assert not bad, bad[:5]
