"""The hand-placed waits of step_hot_kernel, checked on the built code objects (CPU: needs llvm-objdump only)."""
import os

import pytest

import isa_lint

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_the_scan_sees_a_touch_before_the_wait_and_none_after():
    bad = ["global_load_dwordx2 v[2:3], v4, s[0:1]", "v_mov_b64_e32 v[8:9], v[2:3]", "s_waitcnt vmcnt(0)"]
    good = ["global_load_dwordx2 v[2:3], v4, s[0:1]", "v_add_u32_e32 v4, 1, v4", "s_waitcnt vmcnt(0)", "v_mov_b64_e32 v[8:9], v[2:3]"]
    counted = ["global_load_dwordx4 v[2:5], v20, s[0:1]", "global_load_dwordx4 v[6:9], v20, s[0:1]", "s_waitcnt vmcnt(1)", "v_add_f64 v[10:11], v[2:3], v[4:5]",
               "v_add_f64 v[12:13], v[6:7], v[8:9]"]
    overwritten = ["global_load_dwordx2 v[2:3], v4, s[0:1]", "v_mov_b64_e32 v[2:3], 0", "s_waitcnt vmcnt(0)"]
    assert len(isa_lint.lint(bad)) == 1
    assert isa_lint.lint(good) == []
    assert [f[0] for f in isa_lint.lint(counted)] == [4]  # the first load has been waited for, the second has not
    assert len(isa_lint.lint(overwritten)) == 1


@pytest.mark.parametrize("flavour", ["release", "tuning"])
def test_step_hot_kernel_touches_no_register_whose_load_is_in_flight(flavour):
    from hydrochrono_amd import build as hb
    if not os.path.exists(isa_lint.OBJDUMP):
        pytest.skip("llvm-objdump not found")
    co = hb.KERNEL_CO if flavour == "release" else hb.TUNING_CO
    if not os.path.exists(co):
        pytest.skip("code object not built")
    for symbol in ("step_hot_kernelILi1", "step_hot_kernelILi2"):
        lines = isa_lint.kernel_lines(co, symbol)
        assert len(lines) > 500, (symbol, len(lines))
        n_requests = sum(1 for ln in lines if ln.startswith("global_load"))
        assert n_requests >= 4 + 15 + 12 + 12, (symbol, n_requests)  # state, tables, term groups, K words
        findings = isa_lint.lint(lines)
        assert findings == [], (symbol, findings[:5])
