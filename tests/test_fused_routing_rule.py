"""The `auto` routing rule of the multi-link aggregator (csrc/multilink.hip resolve_order2: fused aggregate -> contract kernel
vs the better unfused order) replayed against the table it was fitted from, profiles/r5_fused_routing.txt (one MI355X, per
direction, forward + backward, ms).  VERDICT r5 weak #9 / next #6: no test pinned the rule's DECISIONS.  Host-only: the rule is a
function of the plan's sizes, the widths and the accumulation (sg_multilink_agg_resolve_order2 touches no device memory)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLE = os.path.join(ROOT, "profiles", "r5_fused_routing.txt")
LINE = re.compile(r"^\s*(\d+) x\s+(\d+)\s+(\d+) R\s*(\d+) into (user|movie)\s*:.*?fused\s+([\d.]+)\s+unfused\s+([\d.]+)")


def _rows():
    rows = []
    for line in open(TABLE):
        m = LINE.match(line)
        if m:
            nu, ni, nnz, R = (int(m.group(i)) for i in range(1, 5))
            n_dst, n_src = (nu, ni) if m.group(5) == "user" else (ni, nu)
            rows.append((n_dst, n_src, nnz, R, float(m.group(6)), float(m.group(7))))
    return rows


def _resolve(n_dst, n_src, nnz, R, in_dim=256, units=256, accum=0, order=0):
    from star_gcn_amd import _lib as L
    st = L.MultiLinkPlanStruct()
    st.n_dst, st.n_src, st.nnz, st.num_links = n_dst, n_src, nnz, R
    st.struct_bytes = ctypes.sizeof(L.MultiLinkPlanStruct)
    rc = L.lib().sg_multilink_agg_resolve_order2(ctypes.cast(ctypes.pointer(st), ctypes.c_void_p), order, in_dim, units, accum)
    assert rc >= 0
    return ("auto", "transform_first", "aggregate_first", "fused")[rc]


def test_table_parses():
    rows = _rows()
    assert len(rows) >= 50 and any(r[:2] == (69878, 10677) for r in rows) and any(r[:2] == (1250000, 1000000) for r in rows)


@pytest.mark.skipif(os.environ.get("SG_FUSED") is not None, reason="SG_FUSED overrides the size rule")
def test_auto_rule_agrees_with_the_measured_table():
    """Where the rule picks the fused kernel it must not have lost by more than 3 % in the table; where the fused kernel won by
    15 % or more the rule must pick it; the MovieLens-10M shape (fused 1.5x slower) stays unfused, the config-5 shard fused."""
    picked = lost = 0
    for n_dst, n_src, nnz, R, t_f, t_u in _rows():
        got = _resolve(n_dst, n_src, nnz, R)
        ratio = t_f / t_u
        if got == "fused":
            picked += 1
            assert ratio <= 1.03, ("rule picks fused where it measured slower", n_dst, n_src, R, ratio)
        else:
            assert got == ("transform_first" if n_src <= n_dst else "aggregate_first")
            assert ratio >= 0.85, ("rule leaves a >= 15 % win of the fused kernel unused", n_dst, n_src, R, ratio)
            lost += ratio < 1.0
    assert picked >= 20
    assert _resolve(69878, 10677, 10000006, 10) == "transform_first" and _resolve(10677, 69878, 10000006, 10) == "aggregate_first"
    assert _resolve(1250000, 1000000, 125009847, 16) == "fused" and _resolve(1000000, 1250000, 125009847, 16) == "fused"


def test_auto_never_routes_other_widths_or_stack_to_the_fused_kernel():
    """The kernel handles rows of 4 .. 256 floats, 1 .. 256 units per level and 'stack' (round 6), but it is built and measured for
    256 -> 256 'sum': `auto` keeps every other shape on the unfused orders, an explicit request is honoured."""
    big = (1250000, 1000000, 125009847, 16)
    assert _resolve(*big) == "fused"
    for in_dim, units, accum in [(128, 128, 0), (64, 250, 0), (256, 250, 0), (256, 256, 1), (64, 50, 1)]:
        assert _resolve(*big, in_dim=in_dim, units=units, accum=accum) in ("transform_first", "aggregate_first")
        assert _resolve(*big, in_dim=in_dim, units=units, accum=accum, order=3) == "fused"
