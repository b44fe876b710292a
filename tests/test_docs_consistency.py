"""The measured numbers quoted in DESIGN.md / BASELINE.md / README.md are the ones in profiles/.

Every result table sits in a `<!-- gen:NAME -->` block written by tools/docs_tables.py from the committed evidence
(bench JSON lines, rocprofv3 kernel statistics, the PMC model).  This test regenerates each block and fails when its
text differs from the generator's or a number is more than 10 % away from the evidence; it also checks that the PMC
model bench.py replays was measured on the kernel sources in the tree (round 2 shipped a K4 figure that was 40 % off
its own profile, and a model can go stale the same way)."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)
import docs_tables  # noqa: E402


def _doc(name):
    with open(os.path.join(ROOT, name), encoding="utf-8") as f:
        return f.read()


@pytest.mark.parametrize("name", docs_tables.DOCS)
def test_result_tables_agree_with_profiles(name):
    blocks = docs_tables.blocks_in(_doc(name))
    assert any(b == "headline" for b, _ in blocks), f"{name} quotes no generated headline table"
    for block, body in blocks:
        assert block in docs_tables.BLOCKS, f"{name}: unknown block {block}"
        msg = docs_tables.compare(body, docs_tables.BLOCKS[block]())
        assert msg is None, f"{name} [{block}]: {msg}"


def test_design_has_every_block():
    have = {b for b, _ in docs_tables.blocks_in(_doc("DESIGN.md"))}
    assert have == set(docs_tables.BLOCKS), sorted(set(docs_tables.BLOCKS) - have)


def test_compare_catches_a_stale_number():
    gen = docs_tables.headline()
    assert docs_tables.compare(gen, gen) is None
    first = docs_tables._NUM.findall(gen.split("\n")[2])
    ms = next(x for x in first if "." in x)                        # ms per frame of the c3 row
    stale = gen.replace(f"| {ms} |", f"| {float(ms) * 1.15:.2f} |", 1)
    assert stale != gen and "documented" in docs_tables.compare(stale, gen)
    near = gen.replace(f"| {ms} |", f"| {float(ms) * 1.05:.2f} |", 1)
    assert docs_tables.compare(near, gen) is None                  # inside the 10 % band
    assert "text differs" in docs_tables.compare(gen.replace("Workload", "Work load"), gen)


def test_pmc_model_was_measured_on_these_kernel_sources():
    from vulkanrenderer_amd.fingerprint import kernel_source_fingerprint
    with open(os.path.join(ROOT, "profiles", "r3_k1_valu_model.json")) as f:
        model = json.load(f)
    assert model["source_fingerprint"] == kernel_source_fingerprint(), \
        "kernel sources changed after the last re-profile: run tools/refresh_r3.sh (bench.py would report model_fresh: false)"
    for wl in ("c3", "c4"):
        for k in ("valu_insts_per_ray", "mean_issue_cycles_lower", "td_busy_frac", "hbm_bytes_per_launch", "k2_valu_insts_per_iteration",
                  "k2_mean_issue_cycles_lower", "k2_pixels", "k2_hbm_bytes_per_iteration"):
            assert model[wl][k] > 0, (wl, k)


def test_fingerprint_ignores_comments_only():
    from vulkanrenderer_amd import fingerprint
    a = "int x = 1; // one\n/* block\n comment */ float y;\n"
    assert fingerprint._code_only(a) == fingerprint._code_only("int x = 1;\nfloat   y; // other words\n")
    assert fingerprint._code_only(a) != fingerprint._code_only(a.replace("= 1", "= 2"))
