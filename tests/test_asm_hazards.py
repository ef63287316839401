"""The root cause of the intermittent GPU memory faults of rounds 3-5 as a regression test on the BUILT code objects (no GPU needed).

gfx9-family hardware needs five wait states between a VALU write of an SGPR (v_readlane_b32 restoring a spilled scalar, v_readfirstlane_b32) and a
vector-memory instruction that reads it as scalar base; the compiler cannot see inside the inline assembly of the engine's HBM -> LDS copies and
write-through stores (csrc/pbdx_sweep.hip: lds_dma16, store_pos), which therefore start with `s_nop 4`.  scripts/check_asm_hazards.py disassembles the
gfx950 code objects of every library the suite loads and must find no such site; it must find the site in the listing rocgdb stopped at
(profiles/r05e_*: the range-checked build before the fix)."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("check_asm_hazards", os.path.join(ROOT, "scripts", "check_asm_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_the_checker_finds_the_site_rocgdb_stopped_at():
    tool = _tool()
    # persistent_kernel<8191, 256> of the range-checked build, round 5 before the fix (rocgdb, precise memory: SIGBUS at the store)
    listing = ["s_andn2_saveexec_b64 s[28:29], s[28:29]", "s_cbranch_execz 65415", "v_readlane_b32 s12, v241, 11", "ds_read_b128 v[26:29], v3",
               "v_lshlrev_b32_e32 v5, 4, v23", "v_readlane_b32 s13, v241, 12", "s_waitcnt lgkmcnt(0)", "global_store_dwordx4 v5, v[26:29], s[12:13] sc1"]
    found = tool.scan(listing)
    assert len(found) == 1 and found[0][1] == 1 and "v_readlane_b32 s13" in found[0][3]
    # the same with the guard the sources now carry
    fixed = listing[:-1] + ["s_nop 4", "global_store_dwordx4 v5, v[26:29], s[12:13] sc1"]
    assert tool.scan(fixed) == []
    # a compiler-generated load whose base an s_load wrote is none of this test's business
    assert tool.scan(["s_load_dwordx2 s[4:5], s[0:1], 0x0", "s_waitcnt lgkmcnt(0)", "global_load_dword v1, v0, s[4:5]"]) == []


def test_the_checker_knows_the_one_wait_state_hazards_of_hand_written_memory_instructions():
    """A store of more than 64 bits must not be followed directly by a VALU write of its data registers, nor an HBM -> LDS copy directly by the
    write of M0 it takes its LDS base from: the compiler keeps its own instructions apart, it cannot do so across inline assembly."""
    tool = _tool()
    assert len(tool.scan_other(["global_store_dwordx4 v5, v[26:29], s[12:13] sc1", "v_mov_b32_e32 v27, v1"])) == 1
    assert len(tool.scan_other(["global_store_dwordx3 v[4:5], v[26:28], off", "v_add_f32_e32 v28, v1, v2"])) == 1
    assert len(tool.scan_other(["buffer_store_dwordx4 v[26:29], v4, s[0:3], 0 offen", "v_mul_f32_e32 v26, v1, v2"])) == 1
    assert tool.scan_other(["global_store_dwordx4 v5, v[26:29], s[12:13] sc1", "v_mov_b32_e32 v30, v1"]) == []
    assert tool.scan_other(["global_store_dwordx4 v5, v[26:29], s[12:13] sc1", "s_nop 0", "v_mov_b32_e32 v27, v1"]) == []
    assert tool.scan_other(["global_store_dwordx2 v5, v[26:27], s[12:13]", "v_mov_b32_e32 v27, v1"]) == []       # (64 bits: no hazard)
    assert len(tool.scan_other(["s_mov_b32 m0, s7", "global_load_lds_dwordx4 v3, s[4:5] sc1"])) == 1
    assert tool.scan_other(["s_mov_b32 m0, s7", "s_nop 0", "global_load_lds_dwordx4 v3, s[4:5] sc1"]) == []


@pytest.mark.parametrize("lib", ["libpbdx.so", "libpbdx_bounds.so", "libpbdx_fma.so"])
def test_no_hand_written_memory_instruction_reads_a_freshly_valu_written_scalar(lib):
    path = os.path.join(ROOT, "positionbaseddynamics_amd", "_lib", lib)
    if not os.path.exists(path) or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("library or llvm-objdump not present")
    tool = _tool()
    kernels = tool.disassemble(path)
    assert any("persistent_kernel" in k for k in kernels) and any("fused_kernel" in k for k in kernels)
    sites = [(k, s) for k, ins in kernels.items() for s in tool.scan(ins)]
    assert not sites, "%d hazard site(s), first: %r" % (len(sites), sites[0])
    other = [(k, s) for k, ins in kernels.items() for s in tool.scan_other(ins)]
    assert not other, "%d one-wait-state hazard site(s), first: %r" % (len(other), other[0])
    # every hand-written copy / store is there and guarded
    n_dma = sum(1 for ins in kernels.values() for t in ins if t.startswith("global_load_lds_dwordx4"))
    assert n_dma > 100


@pytest.mark.parametrize("lib", ["libpbdx.so", "libpbdx_fma.so"])
def test_position_scatter_is_a_twelve_byte_lds_store(lib):
    """TileAccess::st stores a projected endpoint as a 3-vector through a 16-byte-aligned pointer and relies on the backend emitting
    ds_write_b96 -- widened to ds_write_b128 it would overwrite the inverse mass kept in .w with an undefined lane.  Checked on the built code
    objects: every sweep kernel scatters with ds_write_b96, and the 16-byte LDS stores it has left are the handful of the fill (integrated
    positions with their inverse mass, the dictionary table), far fewer than its scatters."""
    path = os.path.join(ROOT, "positionbaseddynamics_amd", "_lib", lib)
    if not os.path.exists(path) or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("library or llvm-objdump not present")
    kernels = _tool().disassemble(path)
    sweeps = {k: ins for k, ins in kernels.items() if "persistent_kernel" in k or "fused_kernel" in k}
    assert len(sweeps) >= 20
    for k, ins in sweeps.items():
        b96 = sum(1 for t in ins if t.startswith("ds_write_b96"))
        b128 = sum(1 for t in ins if t.startswith("ds_write_b128"))
        assert b96 >= 8 and b128 <= 16 and b128 * 3 <= b96, (k, b96, b128)
