"""tools/check_v8_asm.py — the build-time guard of the uncounted inline-assembly loads — on synthetic
assembly: it must pass the shape the compiler produces today and fail on the two hazards it exists for."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_v8_asm", os.path.join(ROOT, "tools", "check_v8_asm.py"))
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)

GOOD = """
_Z11k_render_v8ILi0ELi4ELi256ELi24EEvv:
\ts_mov_b64 s[10:11], exec
\t; RT_V8_LOADS_BEGIN (tools/check_v8_asm.py)
\tglobal_load_dwordx4 v[16:19], v[66:67], off
\tglobal_load_dwordx4 v[12:15], v[66:67], off offset:16
\tglobal_load_dwordx4 v[8:11], v[66:67], off offset:32
\tglobal_load_dwordx2 v[44:45], v[66:67], off offset:48
\t; RT_V8_LOADS_END
\tv_mov_b32_e32 v63, v7
\ts_cbranch_execz .LBB0_2
\ts_waitcnt vmcnt(0) ; RT_V8_WAIT_TRI
\tv_mul_f32_e32 v63, v34, v10
.LBB0_2:
\ts_waitcnt vmcnt(0) ; RT_V8_WAIT_PAIR
\tv_pk_add_f32 v[16:17], v[16:17], v[30:31]
\ts_endpgm
  - .name:           _Z11k_render_v8ILi0ELi4ELi256ELi24EEvv
    .sgpr_count:     106
    .sgpr_spill_count: 2
    .vgpr_count:     108
    .vgpr_spill_count: 0
"""


def _run(tmp_path, text):
    p = tmp_path / "k.s"
    p.write_text(text)
    return chk.check(str(p))


def test_todays_shape_passes(tmp_path):
    groups, errors = _run(tmp_path, GOOD)
    assert groups == 1 and errors == []


def test_a_copy_of_an_in_flight_register_is_caught(tmp_path):
    bad = GOOD.replace("\tv_mov_b32_e32 v63, v7\n", "\tv_mov_b32_e32 v63, v12\n")     # reads v12 before any wait
    groups, errors = _run(tmp_path, bad)
    assert groups == 1 and len(errors) == 1 and "in-flight" in errors[0]


def test_spill_traffic_inside_a_load_window_is_caught(tmp_path):
    bad = GOOD.replace("\tv_mov_b32_e32 v63, v7\n", "\tscratch_store_dword off, v63, off offset:16 ; 4-byte Folded Spill\n") \
        .replace(".vgpr_spill_count: 0", ".vgpr_spill_count: 1")
    groups, errors = _run(tmp_path, bad)
    assert any("spill traffic" in e for e in errors)


def test_spills_outside_the_load_windows_are_only_reported(tmp_path, capsys):
    ok = GOOD.replace("\ts_mov_b64 s[10:11], exec\n", "\tscratch_store_dword off, v63, off offset:16 ; 4-byte Folded Spill\n\ts_mov_b64 s[10:11], exec\n") \
        .replace(".vgpr_spill_count: 0", ".vgpr_spill_count: 1")
    groups, errors = _run(tmp_path, ok)
    assert groups == 1 and errors == []
    assert "vector register spills" in capsys.readouterr().out


def test_a_missing_marker_is_caught(tmp_path):
    groups, errors = _run(tmp_path, "_Z11k_render_v8ILi0EEvv:\n\ts_endpgm\n")
    assert groups == 0 and errors


# k_wf_trace with the LDS copy of the top records: the group also issues ds_read into the same registers, and the
# consumers' waits must then cover lgkmcnt as well
WITH_LDS = GOOD.replace("\t; RT_V8_LOADS_END", "\tds_read_b128 v[16:19], v70\n\tds_read_b128 v[12:15], v70 offset:16\n"
                        "\tds_read_b128 v[8:11], v70 offset:32\n\tds_read_b64 v[44:45], v70 offset:48\n\t; RT_V8_LOADS_END") \
    .replace("s_waitcnt vmcnt(0) ;", "s_waitcnt vmcnt(0) lgkmcnt(0) ;")


def test_lds_reads_in_the_group_pass_with_both_counters_waited_for(tmp_path):
    groups, errors = _run(tmp_path, WITH_LDS)
    assert groups == 1 and errors == []


def test_lds_reads_need_the_lgkm_wait(tmp_path):
    bad = WITH_LDS.replace("s_waitcnt vmcnt(0) lgkmcnt(0) ; RT_V8_WAIT_PAIR", "s_waitcnt vmcnt(0) ; RT_V8_WAIT_PAIR")
    groups, errors = _run(tmp_path, bad)
    assert groups == 1 and len(errors) == 1 and "in-flight" in errors[0]
