"""The committed rocprofv3 summaries are what they are named (`r04_bench_kernel_stats.csv` held the
predictor's trace for a whole round because a child run of another config overwrote it).  CPU-only: parses profiles/."""
import csv
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")


def _families(path):
    fams = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Name"]
            m = re.match(r"(?:void )?(?:\(anonymous namespace\)::)?([A-Za-z0-9_]+?)(?:_kernel|_tiled_kernel|_q_kernel)?[<(]", name)
            fam = m.group(1) if m else name
            fams[fam] = fams.get(fam, 0) + int(r["Calls"])
    return fams


def _rounds(kind):
    out = []
    for p in sorted(glob.glob(os.path.join(PROF, f"r*_{kind}_kernel_stats.csv"))):
        m = re.match(r"r(\d+)", os.path.basename(p))
        if m and int(m.group(1)) >= 5:        # the naming is enforced from round 5 on (bench.py: one file per traced config)
            out.append(p)
    return out


def test_the_training_step_trace_is_the_training_step():
    files = _rounds("bench")
    assert files, "no profiles/rNN_bench_kernel_stats.csv of round >= 5 (tools/round_profiles.sh writes it)"
    for p in files:
        fams = _families(p)
        for need in ("pw_wgrad", "bn_bwd_apply", "se_bwd_reduce", "pw_fwd", "conv_wgrad"):
            assert any(k.startswith(need) for k in fams), (os.path.basename(p), need, sorted(fams)[:20])
        # 46 weight-gradient launches per step and 71 apply launches: the ratio identifies the step, whatever the step count
        wg = sum(v for k, v in fams.items() if k.startswith("pw_wgrad"))
        ap = sum(v for k, v in fams.items() if k.startswith("bn_bwd_apply"))
        assert 1.4 < ap / wg < 1.7, (wg, ap)


@pytest.mark.parametrize("kind,absent", [("predict_fbf", ("pw_wgrad", "bn_bwd_apply")), ("long004", ())])
def test_the_other_configs_traces_are_their_own(kind, absent):
    for p in _rounds(kind):
        fams = _families(p)
        assert any(k.startswith("pw_fwd") for k in fams), os.path.basename(p)
        for k in absent:
            assert not any(f.startswith(k) for f in fams), (os.path.basename(p), k)


def test_a_steady_state_predictor_frame_has_no_torch_kernel():
    """the frame-by-frame predictor's glue (index_select x2, index_put, arange, remainder, sigmoid, fills) is
    gone from the steady state - ring / store addressing by mds_copy_rows, sigmoid + TTA mean in head_fwd, zero arenas by
    hipMemsetAsync.  What is left of at::native in the committed trace are one-off set-up launches (a handful of calls)."""
    files = [p for p in _rounds("predict_fbf") if int(re.match(r"r(\d+)", os.path.basename(p)).group(1)) >= 5]
    assert files
    p = files[-1]
    frames = 0
    with open(p) as f:
        rows = list(csv.DictReader(f))
    for r in rows:
        if r["Name"].startswith("stem_fwd") or "stem_fwd_kernel" in r["Name"]:
            frames = max(frames, int(r["Calls"]))
    assert frames >= 100, frames
    assert any("copy_rows_kernel" in r["Name"] and int(r["Calls"]) >= 2 * frames for r in rows)
    for r in rows:
        if "at::native" in r["Name"]:
            assert int(r["Calls"]) < 0.1 * frames, (r["Name"][:80], r["Calls"], frames)
