"""CPU test of build.py's resource guard: the kernels that count their own `s_waitcnt vmcnt(N)` must
compile without register spills / scratch and inside the register budget their blocks-per-CU plan assumes.  The guard runs at
build time; this pins the table and shows that it fires."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def build_mod():
    path = os.path.join(ROOT, "optical-flow-guided-feature-pytorch_amd", "build.py")
    spec = importlib.util.spec_from_file_location("offk_build_guard", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build()                      # no-op when the objects are current
    return mod


def test_guarded_kernels_have_no_spills_and_fit_their_budget(build_mod):
    table = build_mod.resource_table()
    assert set(table) == {"pw_tdiff.hip", "pw_tdiff_split.hip", "chain_fused.hip", "conv_igemm.hip", "wino_gemm.hip"}
    sp = [k for k in table["pw_tdiff_split.hip"] if "pw_tdiff_split_kernel" in k["name"]]
    assert len(sp) == 1 and sp[0]["vgpr_count"] <= 256            # two blocks per CU (LDS); asm loads with counted waits
    wg = [k for k in table["wino_gemm.hip"] if "wino_gemm_kernel" in k["name"]]
    assert len(wg) == 1 and wg[0]["vgpr_count"] <= 128             # four blocks per CU (LDS); it counts its epilogue's stores
    k16 = [k for k in table["pw_tdiff.hip"] if "pw_tdiff16_kernel" in k["name"]]
    assert len(k16) == 1 and k16[0]["vgpr_count"] <= 128          # four blocks per CU
    chains = [k for k in table["chain_fused.hip"] if "chain14_kernel" in k["name"]]
    assert len(chains) == 6 and all(k["vgpr_count"] <= 168 for k in chains)      # three shapes x (direct | Winograd 3x3)
    dflt = [k for k in table["conv_igemm.hip"] if ", 1, 1, 2, 2, 4>" in k["name"]]
    assert len(dflt) == 4 and all(k["vgpr_count"] <= 128 for k in dflt)         # 1x1, 3x3, 5x5 / 2, 7x7 / 2 (four blocks per CU: LDS)
    for rows in table.values():
        for k in rows:
            assert k["vgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0, k
    for k in chains + dflt:          # (pw_tdiff16_kernel parks five loop-invariant scalars in VGPR lanes in front of its K loop)
        assert k["sgpr_spill_count"] == 0, k


def test_guard_fires(build_mod):
    ok = {"name": "offk::pw_tdiff16_kernel(offk::PtParams)", "vgpr_count": 128, "agpr_count": 0, "vgpr_spill_count": 0,
          "sgpr_spill_count": 0, "private_segment_fixed_size": 0}
    assert build_mod.check_resources("pw_tdiff.hip", [ok]) == [ok]
    for bad in ({"vgpr_spill_count": 2, "private_segment_fixed_size": 12}, {"vgpr_count": 136}):
        with pytest.raises(RuntimeError, match="resource guard failed"):
            build_mod.check_resources("pw_tdiff.hip", [dict(ok, **bad)])
    chain = dict(ok, name="void offk::chain14_kernel<4, true>(offk::ChainArgs)", vgpr_count=152)
    assert build_mod.check_resources("chain_fused.hip", [chain]) == [chain]
    with pytest.raises(RuntimeError, match="resource guard failed"):
        build_mod.check_resources("chain_fused.hip", [dict(chain, sgpr_spill_count=3)])
    with pytest.raises(RuntimeError, match="no kernel of pw_tdiff.hip matches"):
        build_mod.check_resources("pw_tdiff.hip", [dict(ok, name="offk::renamed_kernel(offk::PtParams)")])


def test_store_data_hazard_scanner(build_mod):
    """A VALU write to a data register of a 16-byte store directly behind it (MI355X: the new value reached memory in some lanes,
    wino_gemm_split.hip round 5) is refused; a write to another register, a compare, or any instruction in between is not."""
    hit = "  buffer_store_dwordx4 v[42:45], v51, s[56:59], s25 offen   // 000000001A2C\n  v_add_u32_e32 v42, s17, v109\n"
    assert len(build_mod.store_data_hazards(hit)) == 1
    assert len(build_mod.store_data_hazards(hit.replace("v_add_u32_e32 v42", "v_pk_add_f32 v[44:45]"))) == 1
    assert build_mod.store_data_hazards(hit.replace("v_add_u32_e32 v42", "v_add_u32_e32 v46")) == []
    assert build_mod.store_data_hazards(hit.replace("v_add_u32_e32 v42,", "v_cmp_le_i32_e32 vcc,")) == []
    assert build_mod.store_data_hazards(hit.replace("\n  v_add", "\n  s_nop 1\n  v_add")) == []
    assert build_mod.store_data_hazards(hit.replace("dwordx4 v[42:45]", "dwordx2 v[42:43]")) == []
