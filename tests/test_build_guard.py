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
    # one SALU instruction between the store and the VALU write is still inside the window; an `s_nop 1` is the cure
    assert len(build_mod.store_data_hazards(hit.replace("\n  v_add", "\n  s_mov_b32 s4, s5\n  v_add"))) == 1
    assert build_mod.store_data_hazards(hit.replace("\n  v_add", "\n  s_mov_b32 s4, s5\n  s_nop 0\n  v_add")) == []


def test_counted_load_guard_of_the_split_units_kernel(build_mod):
    """pw_tdiff_split_kernel's three counted waits per K-tile step rest on the step issuing exactly 6 register loads + 3 LDS-DMAs, then
    1 LDS-DMA + 3 register loads: the guard accepts that sequence and refuses an extra VMEM instruction, a missing one, and
    a copy of a register an in-flight load writes."""
    wg = ["buffer_load_dwordx4 v[%d:%d], v204, s[44:47], s94 offen" % (52 + 4 * i, 55 + 4 * i) for i in range(6)]
    dma = ["buffer_load_dwordx4 v240, s[52:55], s94 offen lds"] * 4
    wd = ["buffer_load_dwordx4 v[%d:%d], v204, s[44:47], s52 offen" % (36 + 4 * i, 39 + 4 * i) for i in range(3)]
    mf = "v_mfma_f32_16x16x32_bf16 v[188:191], v[12:15], v[218:221], v[188:191]"
    step = ["s_waitcnt vmcnt(7)"] + [x for w in wg for x in (mf, w)] + [x for d in dma[:3] for x in (mf, d)] + ["s_waitcnt vmcnt(9)", mf, dma[3], mf] + wd + ["s_barrier"]
    args = (7, ("r",) * 6 + ("d",) * 3, 9, ("d",) + ("r",) * 3)
    assert build_mod.counted_load_steps(step + step, *args) == 2
    with pytest.raises(RuntimeError, match="VMEM sequence"):
        build_mod.counted_load_steps(step[:5] + ["buffer_load_dwordx4 v[4:7], v1, s[0:3], 0 offen"] + step[5:], *args)
    with pytest.raises(RuntimeError, match="VMEM sequence"):
        build_mod.counted_load_steps([s for s in step if s != wd[0]], *args)
    with pytest.raises(RuntimeError, match="unexpected vector-memory"):
        build_mod.counted_load_steps(step[:5] + ["scratch_store_dword off, v3, s32"] + step[5:], *args)
    with pytest.raises(RuntimeError, match="in-flight load destination"):
        build_mod.counted_load_steps(step[:4] + ["v_mov_b32_e32 v9, v53"] + step[4:], *args)


def test_store_data_hazard_scanner_reads_the_data_operand_of_global_stores(build_mod):
    g = "  global_store_dwordx4 v[18:19], v[52:55], off offset:192\n  v_pk_add_f32 v[18:19], v[122:123], v[146:147]\n"
    assert build_mod.store_data_hazards(g) == []                      # the ADDRESS pair is free once the store has issued
    assert len(build_mod.store_data_hazards(g.replace("v_pk_add_f32 v[18:19]", "v_pk_add_f32 v[54:55]"))) == 1
