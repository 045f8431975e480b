"""No 32x32 MFMA of the translation units built with -amdgpu-mfma-vgpr-form may write its result over its own SrcA / SrcB
(csrc/mfma_guard.h: the compiler's untied VGPR form of a zero-initialised MFMA lacks the early-clobber; the hardware then
computes some samples of a tile with half-read operands when the matrix pipe is contended -- intermittent, small, and
invisible to every test that runs one workgroup per CU).  The device code is compiled to assembly here (hipcc cross-compiles
without a GPU) and scanned by tools/mfma_overlap.py."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


# (mlp32.hip -- the exact-fp32 MFMA unit -- is built in the default AGPR form, whose MFMAs carry the early-clobber; it is
#  scanned all the same: at low register pressure the compiler picks the VGPR form by itself)
@pytest.mark.parametrize("src", ["nerf_mlp.hip", "nerf_mlp_bwd.hip", "mlp32s.hip", "mlp32s_f16.hip", "ffmlp.hip", "mlp32.hip"])
def test_no_mfma_result_lands_on_its_operands(src, tmp_path):
    from enerf_amd import build
    flags = [f for f in build.FLAGS if f not in ("-fPIC",)] + build.EXTRA.get(src, [])
    if src != "mlp32.hip":
        assert "-amdgpu-mfma-vgpr-form" in flags, f"{src} is no longer built in the VGPR form: drop it from this test"
    out = tmp_path / (src + ".s")
    subprocess.check_call([build._hipcc()] + flags + ["-S", "--cuda-device-only", os.path.join(build.CSRC, src), "-o", str(out)],
                          stderr=subprocess.DEVNULL)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "mfma_overlap.py"), str(out)], capture_output=True,
                         text=True, check=True).stdout
    rows = [ln for ln in res.splitlines() if " mfma " in ln]
    assert rows, "no MFMA found: the scan is looking at the wrong thing"
    bad = [ln for ln in rows if not ln.rstrip().endswith("overlapping 0")]
    assert not bad, "MFMA results overlapping SrcA / SrcB:\n" + "\n".join(bad[:10])


# kernels that run two or more wavefronts per SIMD (nerf_mlp_bwd.hip runs one, by construction, and is compiled without the
# barrier on purpose: see that file)
@pytest.mark.parametrize("src", ["nerf_mlp.hip", "mlp32s.hip", "mlp32s_f16.hip", "ffmlp.hip", "ffnerf.hip", "mlp32.hip"])
def test_no_conversion_writes_an_operand_right_in_front_of_its_mfma(src, tmp_path):
    """csrc/mlp32s_ops.h (operand_ready): a v_cvt_pk_* / v_pk_add_f32 that writes an MFMA's SrcA / SrcB must not sit within
    four issue slots of it -- on gfx950 the matrix pipe otherwise reads lanes 16..31 / 48..63 of the operand before the
    conversion has written them once several wavefronts share the SIMD.  Scanned on the compiled assembly by
    tools/mfma_operand_distance.py."""
    from enerf_amd import build
    import importlib.util
    spec = importlib.util.spec_from_file_location("mfma_operand_distance", os.path.join(ROOT, "tools", "mfma_operand_distance.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    flags = [f for f in build.FLAGS if f not in ("-fPIC",)] + build.EXTRA.get(src, [])
    out = tmp_path / (src + ".s")
    subprocess.check_call([build._hipcc()] + flags + ["-S", "--cuda-device-only", os.path.join(build.CSRC, src), "-o", str(out)],
                          stderr=subprocess.DEVNULL)
    mfmas, close = 0, []
    for name, ins in mod.functions(str(out)).items():
        mfmas += sum(1 for s in ins if s.startswith("v_mfma"))
        close += [(name[:60], d, w) for d, w, _ in mod.scan(ins)
                  if d < 4 and w.startswith(("v_cvt_pk", "v_pk_add_f32", "v_cvt_"))]
    assert mfmas, "no MFMA found: the scan is looking at the wrong thing"
    assert not close, f"conversions within 4 slots of the MFMA that reads them: {close[:8]}"


def test_backward_without_the_operand_barrier_has_a_cu_to_itself(tmp_path):
    """nerf_mlp_bwd.hip compiles k_nerf_bwd WITHOUT the operand barrier (its conversions sit two slots in front of their
    MFMAs: the scan above would flag twenty of them).  That is sound only while one workgroup of it fits a CU, i.e. one
    wavefront per SIMD -- the hazard needs two wavefronts' conversions on one SIMD.  What guarantees it is the kernel's LDS
    claim (more than half of the CU's 160 KiB; a static_assert in the source says the same): read here off the compiled
    kernel descriptor, together with the workgroup size (256 threads = one wavefront per SIMD)."""
    import re
    from enerf_amd import build
    src = "nerf_mlp_bwd.hip"
    flags = [f for f in build.FLAGS if f not in ("-fPIC",)] + build.EXTRA.get(src, [])
    out = tmp_path / (src + ".s")
    subprocess.check_call([build._hipcc()] + flags + ["-S", "--cuda-device-only", os.path.join(build.CSRC, src), "-o", str(out)],
                          stderr=subprocess.DEVNULL)
    text = out.read_text()
    lds = [int(v) for name, v in re.findall(r"\.amdhsa_kernel (\S*k_nerf_bwd\S*)[\s\S]*?\.amdhsa_group_segment_fixed_size (\d+)", text)]
    assert lds, "k_nerf_bwd not found in the unit's assembly"
    assert all(v > 80 * 1024 for v in lds), lds
    # (the metadata block lists .max_flat_workgroup_size a few lines above the kernel's .name)
    flat = [int(m.group(1)) for m in re.finditer(r"\.max_flat_workgroup_size:\s+(\d+)\s*\n\s*\.name:\s+\S*k_nerf_bwdE", text)]
    assert flat and all(v == 256 for v in flat), flat
