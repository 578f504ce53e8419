"""Host-side mirror of the reference's model API (no GPU needed)."""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT

from nano_vs_slam_amd.kp2dtiny.models import kp2dtiny as K
from nano_vs_slam_amd.sharding import shard_range
from oracle import kp2d_oracle as orc
from oracle.weights import spread_state_dict, synthetic_frames

# measured on the imported reference
PARAMS = {("S", False, 28): 928079, ("N", False, 28): 528959, ("S_A", False, 28): 968271,
          ("S", True, 19): 704934, ("S_A", True, 28): 747727}


@pytest.mark.parametrize("key", list(PARAMS))
def test_state_dict_layout_and_param_count(key):
    name, v3, ncls = key
    m = K.tiny_factory(name, ncls, v3=v3)
    mine = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    assert mine == list(orc.state_dict_shapes(orc.get_config(name, v3), ncls).items())
    assert sum(p.numel() for p in m.parameters()) == PARAMS[key]
    assert m.training is True            # reference force-sets it (kp2dtiny.py:456,813)
    m.eval()
    assert m.training is False


def test_tests_py_equivalent_construction():
    """reference tests.py: tiny_factory("S_A", 28, v3=True) builds; attributes callers read exist."""
    from src.kp2dtiny.models.kp2dtiny import tiny_factory   # the reference's import line
    m = tiny_factory("S_A", 28, v3=True)
    assert (m.nfeatures, m.nClasses, m.cell, m.global_desc_dim) == (32, 28, 4, 4096)
    assert m.get_global_desc_dim() == m.get_netvlad_dim() == 4096 and m.get_num_clusters() == 64
    info = m.gather_info()
    assert info["total_params"] == 747727 and info["netvlad_dim"] == 4096
    m.device = "cuda"                      # callers assign it (eval_multitask.py:197-198)
    m.freeze_backbone()
    assert not any(p.requires_grad for p in m.backbone.parameters())
    m.freeze_segmentation(except_last_layer=True)
    assert all(p.requires_grad for p in m.seg_head.convs[-1].parameters())


def test_get_config_contract():
    with pytest.raises(ValueError):
        K.get_config("nope")
    with pytest.raises(ValueError):
        K.get_config("F", v3=True)
    a = K.get_config("S", to_mcu=True)
    assert a["upscale_method"] == "convtranspose" and a["leaky_relu"] is False
    # the reference leaks that mutation into later calls (App. B.20); this build returns copies
    assert "upscale_method" not in K.get_config("S")
    assert K.get_config("N")["num_clusters"] == 32 and "num_clusters" not in K.get_config("N", v3=True)
    assert set(K.KP2DTINY_CONFIGS) == {"S", "S_A", "N", "N_A", "D", "F", "GEM_N", "GEM_S_A", "CONVAP_S_A"}
    assert set(K.KP2DTINYV3_CONFIGS) == {"S", "S_A", "N", "N_A", "D", "D_A", "CONVAP_S_A"}


def test_partial_and_strict_loads():
    m = K.tiny_factory("S", 28)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in
          spread_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}).items()}
    m.load_state_dict(sd, strict=True)
    # callers surgically drop the class layer (train_multitask.py:309-325) and load non-strict
    part = {k: v for k, v in sd.items() if not k.startswith("seg_head.convs.8")}
    res = m.load_state_dict(part, strict=False)
    assert sorted(res.missing_keys) == ["seg_head.convs.8.bias", "seg_head.convs.8.weight"]
    with pytest.raises(RuntimeError):
        m.load_state_dict(part, strict=True)


def test_no_cpu_path():
    m = K.tiny_factory("S", 28).eval()
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.zeros(1, 3, 32, 32))
    with pytest.raises(RuntimeError):
        m.post_processing({"score": torch.zeros(1, 1, 8, 8), "coord": torch.zeros(1, 2, 8, 8),
                           "feat": torch.zeros(1, 32, 16, 16), "seg": torch.zeros(1, 28, 16, 16)}, 32, 32)
    with pytest.raises(RuntimeError):
        m.backbone(torch.zeros(1, 3, 32, 32))      # parameter holders never execute torch ops


def test_batch_stream_has_no_cpu_path():
    """pipeline.BatchStream (batches in flight on HIP streams) refuses a CPU device instead of running anything there."""
    from nano_vs_slam_amd.pipeline import BatchStream
    with pytest.raises(RuntimeError, match="no CPU path"):
        BatchStream(object(), slots=2, device="cpu")


def test_unbuilt_variants_raise_not_fallback():
    with pytest.raises(NotImplementedError):       # heads.py:58 / segmentation.py:120
        K.KP2DTinyV2(**K.get_config("S"), nClasses=28, upscale_method="bilinear")
    K.KP2DTinyV2(**K.get_config("S"), nClasses=28, depth=True)._check_built()      # depth heads are built
    d = K.KP2DTinyV3(**K.get_config("S", v3=True), nClasses=19, depth=True)
    assert d.seg_head.convs[7].conv.weight.shape[0] == 96 and d.seg_head.featD.bias is None


@pytest.mark.parametrize("name,v3", [("S", False), ("S_A", False), ("N", True), ("S_A", True)])
def test_to_mcu_state_dict_layout(name, v3):
    """to_mcu=True: TransposedConvUpsampleModel parameters in the reference's registration order, ReLU."""
    m = K.tiny_factory(name, 28, to_mcu=True, v3=v3)
    assert m.upscale_method == "convtranspose" and m.leaky_relu is False
    mine = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    assert mine == list(orc.state_dict_shapes(orc.get_config(name + "+mcu", v3), 28).items())
    assert K.get_config(name, v3=v3).get("upscale_method", "pixelshuffle") == "pixelshuffle"   # table not mutated


@pytest.mark.parametrize("name,v3", [("D", False), ("F", False), ("D", True), ("D_A", True)])
def test_large_and_tiny_f_state_dict_layout(name, v3):
    m = K.tiny_factory(name, 28, v3=v3)
    mine = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    assert mine == list(orc.state_dict_shapes(orc.get_config(name, v3), 28).items())


@pytest.mark.parametrize("name,v3", [("GEM_S_A", False), ("GEM_N", False), ("CONVAP_S_A", False), ("CONVAP_S_A", True)])
def test_pooler_variants_state_dict_layout(name, v3):
    m = K.tiny_factory(name, 28, v3=v3)
    mine = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    assert mine == list(orc.state_dict_shapes(orc.get_config(name, v3), 28).items())
    assert m.global_desc_dim == m.encoder_dim * 16
    m._check_built()
    e = K.tiny_factory("S", 28, to_export=True)          # remove_netvlad: no pooler parameters at all
    assert not any(k.startswith("vlad_head.netvlad") for k in e.state_dict()) and e.global_desc_dim == 0


def test_shard_range_partitions_frames():
    for n in (1, 7, 64, 256, 257):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 4, 4)


def test_seeded_inputs_are_reproducible():
    a = spread_state_dict({"backbone.conv1a.conv.weight": (16, 3, 3, 3), "backbone.conv1a.bn.running_var": (16,)})
    b = spread_state_dict({"backbone.conv1a.bn.running_var": (16,), "backbone.conv1a.conv.weight": (16, 3, 3, 3)})
    assert all(np.array_equal(a[k], b[k]) for k in a)       # independent of enumeration order
    assert a["backbone.conv1a.bn.running_var"].min() >= 0.5
    x = synthetic_frames(2, 16, 24, seed=7)
    assert x.shape == (2, 3, 16, 24) and x.dtype == np.float32 and -1 <= x.min() and x.max() <= 1
    assert np.array_equal(x, synthetic_frames(2, 16, 24, seed=7))


def test_init_netvlad_matches_reference_fixture():
    """Host-side NetVLAD.init_params (no GPU involved): alpha, centroids and soft-assignment weights as the reference."""
    from conftest import load_golden
    meta, z = load_golden("v2_N_32x48_taps")
    m = K.tiny_factory("N", 28)
    m.init_netvlad(z["init_clsts"].copy(), z["init_descs"].copy())
    nv = m.vlad_head.netvlad
    assert abs(nv.alpha - float(z["init_alpha"])) < 1e-9 * abs(nv.alpha)
    assert np.array_equal(nv.conv.weight.detach().numpy(), z["init_conv_weight"])
    assert np.array_equal(nv.centroids.detach().numpy(), z["init_centroids"])
    assert [k for k in m.state_dict() if "netvlad" in k] == ["vlad_head.netvlad.centroids", "vlad_head.netvlad.conv.weight"]


def test_lightglue_host_mirror_layout_and_no_cpu_path():
    """LightGlue host mirror: reference import lines, state_dict layout, config table; no CPU execution."""
    from lightglue.lightglue import LightGlue
    from lightglue.lightglue_configs import LIGHT_GLUE_CONFIGS, get_light_glue_config
    from oracle import lightglue_oracle as lgo
    assert set(LIGHT_GLUE_CONFIGS) == {"S", "F", "A"}
    with pytest.raises(ValueError):
        get_light_glue_config("Z")
    for name in LIGHT_GLUE_CONFIGS:
        m = LightGlue(get_light_glue_config(name))
        assert [(k, tuple(v.shape)) for k, v in m.state_dict().items()] == \
            list(lgo.state_dict_shapes(lgo.get_config(name)).items())
    m = LightGlue(get_light_glue_config("S")).eval()
    assert m.conf.n_layers == 4 and m.conf["filter_threshold"] == 0.0 and m.required_data_keys[0] == "keypoints0"
    data = {"keypoints0": torch.zeros(1, 8, 2), "keypoints1": torch.zeros(1, 8, 2),
            "descriptors0": torch.zeros(1, 8, 32), "descriptors1": torch.zeros(1, 8, 32)}
    with pytest.raises(RuntimeError):
        m(data)                                      # CPU tensors: refused, never a fallback
    with pytest.raises(RuntimeError):
        m.transformers[0].self_attn(torch.zeros(1, 8, 32))


def test_warp_specialised_conv_kernels_compile_without_spills_and_keep_their_counted_wait(tmp_path):
    """conv3x3_wsm.hip rests on two properties of the generated code that no run-time test sees directly:
    (1) no VGPR spills — a spill inside the matrix phase costs more than the form gains, and scratch traffic of the
        staging waves would join the vmcnt queue;
    (2) the staging loop's counted wait: the LDS-DMA pieces of a weight slab are issued BEFORE the ten loads of the
        next request and `s_waitcnt vmcnt(10)` stands in front of the step's barrier (newer operations only make that
        wait stronger; fewer than ten newer ones would let a barrier pass with a slab still in flight).
    Cross-compiles the file for gfx950 (no GPU needed) and reads the assembly."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    src = os.path.join(ROOT, "nano-vs-slam_amd", "csrc", "conv3x3_wsm.hip")
    out = tmp_path / "wsm.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                    "-I" + os.path.join(ROOT, "nano-vs-slam_amd", "csrc"), "-I" + os.path.join(ROOT, "include"), src, "-o", str(out)],
                   check=True, capture_output=True, timeout=600)
    asm = out.read_text()
    spills = [int(x) for x in re.findall(r"\.vgpr_spill_count:\s+(\d+)", asm)]
    vgprs = [int(x) for x in re.findall(r"\.vgpr_count:\s+(\d+)", asm)]
    # one instantiation per store mode and item width (8), the S16P-input forms (5: NHWC, pooled, S16P, S16P pixel-shuffled, the
    # merged first layer's mix) and the fp32-input form with an S16P pixel-shuffled output
    assert len(spills) == 14 and all(s == 0 for s in spills), spills
    assert all(v <= 168 for v in vgprs), vgprs                               # three waves per SIMD (768 threads per CU)
    kernels = asm.split("s_endpgm")
    checked = checked16 = 0
    for body in kernels:
        if "buffer_load_dwordx4" not in body or " lds" not in body:
            continue
        lines = [ln.strip() for ln in body.splitlines()]
        lines = [ln for ln in lines if ln and not ln.startswith(";")]
        wide = [ln for ln in lines if ln.startswith("buffer_load_dwordx4")]
        if all(ln.endswith("lds") for ln in wide):
            # S16P inputs: the staging waves issue LDS-DMA only (9 weight + 11 image copies per wave and step, at two sites:
            # before the loop and inside it), wait for all of it and pass a BARE barrier — a fence there would be harmless
            # for this role but a vmcnt(0) in front of the multiplying waves' barrier would make them wait for their stores
            assert len(wide) >= 40, len(wide)
            drains = [i for i, ln in enumerate(lines) if ln.startswith("s_waitcnt vmcnt(0)") and
                      any(x.startswith("s_barrier") for x in lines[i + 1:i + 5])]
            assert len(drains) >= 2, len(drains)
            checked16 += 1
            continue
        waits = [i for i, ln in enumerate(lines) if ln.startswith("s_waitcnt vmcnt(10)")]
        assert len(waits) >= 3, len(waits)                                   # prologue + the two half-steps of the loop
        for w in waits[1:]:
            # walking back from the wait: ten plain loads, then (further back) the nine LDS-DMA pieces, nothing else of VMEM
            back = [ln for ln in lines[:w] if ln.startswith("buffer_") or ln.startswith("global_") or ln.startswith("scratch_")]
            tail = back[-14:]
            assert all("lds" not in ln for ln in tail[-10:]), tail[-10:]
            assert all(ln.endswith("lds") for ln in tail[:4]), tail[:4]     # (nine pieces per wave at 64 channels, five / four at 32)
        checked += 1
    assert checked == 9 and checked16 == 5, (checked, checked16)


def test_no_wide_buffer_store_carries_its_offset_in_an_sgpr(tmp_path):
    """The gfx950 store hazard of DESIGN.md §4 (profiles/r4_wsm_store_hazard.txt): a `buffer_store_dwordx{2,3,4}` whose
    `soffset` operand is an SGPR gets no wait state from hipcc before the next VALU write of its data registers (LLVM's
    hazard rule exempts that form), and on gfx950 such a store now and again wrote the NEXT tile's values.  The fix is a
    code shape (scalar part of the address folded into the VGPR offset, soffset = 0), so this test reads the shipped
    code objects: every multi-dword buffer store of every kernel in libkp2d_hip.so must have a literal soffset.
    Disassembles the in-tree library (no GPU, no recompilation)."""
    import re
    import shutil
    import subprocess
    from nano_vs_slam_amd import _lib
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not (os.path.exists(objdump) and os.path.exists(_lib.LIB_PATH)):
        pytest.skip("llvm-objdump or libkp2d_hip.so not found")
    so = tmp_path / "libkp2d_hip.so"
    shutil.copy(_lib.LIB_PATH, so)                                # (--offloading unbundles next to its input)
    subprocess.run([objdump, "--offloading", so.name], check=True, capture_output=True, cwd=tmp_path, timeout=300)
    bundles = sorted(p for p in os.listdir(tmp_path) if "amdgcn" in p and "gfx950" in p)
    assert bundles, os.listdir(tmp_path)
    stores = bad = 0
    for b in bundles:
        asm = subprocess.run([objdump, "-d", b], check=True, capture_output=True, cwd=tmp_path, timeout=300, text=True).stdout
        for ln in asm.splitlines():
            m = re.search(r"\bbuffer_store_dwordx[234]\s+([^/]*)", ln)
            if not m:
                continue
            ops = [o.strip() for o in m.group(1).split(",")]
            # vdata, vaddr | off, srsrc, soffset [offen] [offset:N] ...
            assert len(ops) >= 4, ln
            soffset = ops[3].split()[0]
            stores += 1
            if not re.fullmatch(r"-?\d+|0x[0-9a-fA-F]+", soffset):
                bad += 1
                print("SGPR soffset:", ln.strip())
    assert stores > 300, stores                                   # the conv epilogues alone hold several hundred
    assert bad == 0, f"{bad} of {stores} wide buffer stores carry an SGPR soffset"


def test_lds_dma_kernels_compile_without_spills_and_keep_their_copies_in_flight_across_barriers(tmp_path):
    """conv3x3_s16.hip and mff_tail.hip keep LDS-DMA copies in flight ACROSS workgroup barriers (two steps / slices ahead): that rests on
    (1) bare s_barrier instructions behind a COUNTED s_waitcnt — a `__syncthreads()` there makes hipcc drain vmcnt to 0 because
        of the fence it carries, and the pipeline silently degrades to one step ahead (seen while writing the kernel);
    (2) no VGPR spills (scratch traffic would join the vmcnt queue the counted waits count).
    Cross-compiles both files for gfx950 (no GPU needed) and reads the assembly."""
    import re
    import shutil
    import subprocess
    from concurrent.futures import ThreadPoolExecutor
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    csrc = os.path.join(ROOT, "nano-vs-slam_amd", "csrc")

    def asm_of(name):
        out = tmp_path / (name + ".s")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-I" + csrc,
                        "-I" + os.path.join(ROOT, "include"), os.path.join(csrc, name), "-o", str(out)],
                       check=True, capture_output=True, timeout=900)
        return out.read_text()

    with ThreadPoolExecutor(max_workers=2) as ex:
        s16, mff = ex.map(asm_of, ["conv3x3_s16.hip", "mff_tail.hip"])
    for asm, nk, vmax in ((s16, 7, 168), (mff, 1, 256)):      # (s16: four fp32 / S16P store modes, conv3b's S16P full + pooled, the planar form, the tap kernel)
        spills = [int(x) for x in re.findall(r"\.vgpr_spill_count:\s+(\d+)", asm)]
        vgprs = [int(x) for x in re.findall(r"\.vgpr_count:\s+(\d+)", asm)]
        assert len(spills) == nk and all(v == 0 for v in spills), spills
        assert all(v <= vmax for v in vgprs), vgprs

    def waits_before_barriers(body):
        lines = [ln.strip() for ln in body.splitlines()]
        out = []
        for i, ln in enumerate(lines):
            if ln.startswith("s_barrier"):
                back = [x for x in lines[max(0, i - 8):i] if x.startswith("s_waitcnt")]
                out.append(back[-1] if back else "")
        return out

    # conv3x3_s16.hip, 32-channel items (three image stages): the DMA loop's barrier stands behind vmcnt(11), not vmcnt(0)
    body = next(b for b in s16.split("s_endpgm") if "conv3x3_f16x3_s16_kernelILi5ELi2E" in b)
    w = waits_before_barriers(body)
    assert any(x.startswith("s_waitcnt vmcnt(11)") for x in w), w
    # mff_tail.hip: every barrier of the slice loop behind vmcnt(3) lgkmcnt(0) (or the drain of the last slices), never a plain fence
    w = waits_before_barriers(mff)
    assert sum(1 for x in w if "vmcnt(3)" in x and "lgkmcnt(0)" in x) >= 2, w
