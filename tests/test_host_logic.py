"""Host-side logic that needs no GPU: the StepPlan cache policy and the environment-switch defaults documented in DESIGN.md."""
import os
import re

from chameleon_recsys_amd.nar import nar_model


class _FakePlan:
    def __init__(self, rt, B, T, N, n_buf, Bg):
        self.key = (B, T, N, n_buf, Bg)
        self._bytes = nar_model.StepPlan.estimate_bytes(rt.layout, B, T, N, rt)

    def allocated_bytes(self):
        return self._bytes


def _runtime(max_plans, budget, monkeypatch):
    rt = object.__new__(nar_model.NARRuntime)            # no device: only the cache fields plan() touches
    rt._plans, rt.max_plans, rt.plan_bytes_budget = {}, max_plans, budget
    rt.p3 = rt.h2 = False

    class _L:
        C = 1024
    rt.layout = _L()
    monkeypatch.setattr(nar_model, "StepPlan", type("StepPlan", (_FakePlan,), {
        "estimate_bytes": staticmethod(nar_model.StepPlan.estimate_bytes)}))
    return rt


def test_plan_cache_evicts_least_recently_used_shape(monkeypatch):
    rt = _runtime(3, 1 << 60, monkeypatch)
    a, b, c = rt.plan(256, 10, 50, 3000), rt.plan(256, 11, 50, 3000), rt.plan(256, 12, 50, 3000)
    assert rt.plan(256, 10, 50, 3000) is a                  # a becomes most recently used
    d = rt.plan(256, 13, 50, 3000)                          # evicts b, not everything
    assert [p.key[1] for p in rt._plans.values()] == [12, 10, 13]
    assert rt.plan(256, 12, 50, 3000) is c and rt.plan(256, 13, 50, 3000) is d
    assert rt.plan(256, 11, 50, 3000) is not b              # b was dropped and is rebuilt


def test_warm_plans_builds_every_padded_length_once(monkeypatch):
    """NARRuntime.warm_plans (round 5): a trainer's warm-up builds the buffer set of every padded length T <= truncate_session_length - 1, so
    that no step of the run allocates (bench.py's boundary legs: step_plans_created_in_timed_steps == 0); within the cache's own limits."""
    rt = _runtime(24, 1 << 60, monkeypatch)
    rt.warm_plans(256, range(1, 20), 50, 3000)
    assert [p.key[1] for p in rt._plans.values()] == list(range(1, 20)) and rt.plans_created == 19
    first = dict(rt._plans)
    rt.warm_plans(256, range(1, 20), 50, 3000)                            # a second call finds them all
    assert rt.plans_created == 19 and all(rt._plans[k] is v for k, v in first.items())
    rt2 = _runtime(4, 1 << 60, monkeypatch)
    rt2.warm_plans(256, range(1, 20), 50, 3000)                          # more lengths than the cache holds: the last ones stay
    assert [p.key[1] for p in rt2._plans.values()] == [16, 17, 18, 19]


def test_plan_cache_respects_byte_budget(monkeypatch):
    one = nar_model.StepPlan.estimate_bytes(type("L", (), {"C": 1024}), 256, 19, 50)
    assert 4.0e9 < one < 6.0e9                              # G1 shape: ~4.4 GB of CAR / scorer activations
    # the plane-resident operands of the candidate-row CAR GEMMs are part of the estimate: Z1 and dZ2 as two fp16 /
    # three bf16 planes
    class _Lib:
        @staticmethod
        def cham_group_rows_segments_len(n):
            return n
    planes = lambda h2: nar_model.StepPlan.estimate_bytes(type("L", (), {"C": 1024}), 256, 19, 50,
                                                          type("RT", (), {"p3": True, "h2": h2, "lib": _Lib})) - one
    assert 2.0e9 < planes(True) < 2.2e9 and 3.0e9 < planes(False) < 3.2e9
    rt = _runtime(24, int(2.5 * one), monkeypatch)
    for T in (19, 18, 17, 16):
        rt.plan(256, T, 50, 3000)
    assert 1 <= len(rt._plans) <= 2 and list(rt._plans.values())[-1].key[1] == 16


def test_documented_switch_defaults_match_the_code():
    src = open(nar_model.__file__).read()
    doc = open(nar_model.__file__.rsplit("/chameleon_recsys_amd/", 1)[0] + "/DESIGN.md").read()
    for name, default in re.findall(r'os\.environ\.get\("(CHAM_[A-Z0-9_]+)",\s*"([^"]*)"\)', src):
        assert name in doc, "%s is not documented in DESIGN.md section 9" % name
        row = [l for l in doc.splitlines() if l.startswith("| `%s`" % name)]            # the row OF the switch in section 9's table
        assert row, name
        if name in ("CHAM_COMPACT", "CHAM_OVERLAP", "CHAM_PRESAMPLE", "CHAM_GEMM_H2", "CHAM_GEMM_P3"):
            assert "| %s |" % default in row[0], (name, default, row[0])


def test_roctx_ranges_are_noops_unless_asked_for():
    """chameleon_recsys_amd/_roctx.py: without CHAM_ROCTX the stage ranges of a step cost nothing and load nothing; with CHAM_ROCTX=1 they go through
    the ROCm tools extension library (rocprofv3 --marker-trace shows them)."""
    import subprocess
    import sys
    from chameleon_recsys_amd import _roctx
    if os.environ.get("CHAM_ROCTX", "0") != "1":
        assert not _roctx.enabled() and _roctx.push("x") == -1 and _roctx.pop() == -1
        with _roctx.range_("y"):
            pass
    code = ("import os; os.environ['CHAM_ROCTX'] = '1'\n"
            "from chameleon_recsys_amd import _roctx\n"
            "print(int(_roctx.enabled()), _roctx.push('a') if _roctx.enabled() else 0, _roctx.pop() if _roctx.enabled() else 0)\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stderr
    on, a, b = (int(x) for x in r.stdout.split())
    if os.path.exists("/opt/rocm/lib/libroctx64.so") or os.path.exists("/opt/rocm/lib/librocprofiler-sdk-roctx.so"):
        assert on == 1 and a >= 0 and b >= 0, r.stdout
