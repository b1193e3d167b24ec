"""No kernel of the SHIPPED library may touch scratch memory or spill vector registers (CPU test: reads the gfx950 code objects inside
v2v_amd/libv2v_hip.so -- their amdhsa notes and their disassembly -- through tools/kernel_resources.py; no GPU, no recompilation).

Round 4 shipped `esim_voxel_kernel<float32, 1 pixel, SUM, device noise>` with 560 bytes of scratch and 2,262 scratch instructions (a
select chain over by-reference lambda captures had become a run-time index into the closure object), while the build summary said
"0 spilled VGPRs": `.vgpr_spill_count` does not see that kind of scratch.  This test looks at all three figures."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "v2v_amd", "libv2v_hip.so")

# kernels allowed to use scratch / spill VGPRs: {substring of the demangled name: reason}.  Empty on purpose -- add an entry only with a
# measurement that shows the spilling variant is the faster one.
ALLOW = {}


@pytest.fixture(scope="module")
def resources():
    if not os.path.exists(SO):
        import __graft_entry__
        __graft_entry__.build()
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "tools", "kernel_resources.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.from_so(SO)


def test_every_code_object_is_gfx950_and_the_kernel_families_are_all_there(resources):
    assert len(resources) > 200
    assert {r["arch"] for r in resources.values()} == {"gfx950"}
    names = " ".join(r["demangled"] for r in resources.values())
    for family in ("esim_voxel_kernel", "v2e_voxel_kernel", "frontend_tile_kernel", "convlstm_step_kernel", "events_to_voxel", "count_pick_kernel"):
        assert family in names, family


def test_no_kernel_uses_scratch_or_spills_vector_registers(resources):
    bad = []
    for r in resources.values():
        if any(key in r["demangled"] for key in ALLOW):
            continue
        if r.get("scratch_bytes", 0) or r.get("vgpr_spill", 0) or r["scratch_instructions"]:
            bad.append(f'{r["demangled"]}: scratch {r.get("scratch_bytes", 0)} B, {r.get("vgpr_spill", 0)} spilled VGPRs, {r["scratch_instructions"]} scratch instructions')
    assert not bad, "\n".join(bad)


def test_the_single_clip_float32_instance_of_the_drop_in_emulator_is_scratch_free(resources):
    """EventEmulator.video_to_voxel(float32 clip) (data/v2v_core_esim.py:26-69) launches <float32, 1 pixel, SUM, Philox noise, float32 grid>."""
    hit = [r for r in resources.values() if "esim_voxel_kernel<1, 1, 0, 1, true, false, false, false, false>" in r["demangled"]]
    assert len(hit) == 1
    assert hit[0].get("scratch_bytes", 0) == 0 and hit[0]["scratch_instructions"] == 0 and hit[0]["vgpr"] <= 128


def test_no_kernel_uses_packed_float32_instructions(resources):
    """Round 5 (tools/pk_cohazard_probe.py): packed float32 instructions gave wrong values in lanes 48-63 whenever a matrix-core kernel of
    another stream shared the CU -- in the x2 upsampling kernel and in the v2e simulator.  The library is built with
    `-target-feature -packed-fp32-ops`; no v_pk_mul/add/fma_f32 may come back (an explicit builtin, a new translation unit without the flag)."""
    bad = {r["demangled"][:120]: r["packed_f32_instructions"] for r in resources.values() if r["packed_f32_instructions"]}
    assert not bad, bad
