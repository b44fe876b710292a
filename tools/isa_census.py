#!/usr/bin/env python3
"""Static instruction census of one kernel in a hipcc --save-temps .s file, VALU opcodes priced by the issue classes
tools/calib/issue_rates*.hip measured on gfx950 (cycles per wave-instruction per SIMD at full occupancy):
    fast  2.4-2.8  v_fma/fmac/add/sub/mul f32, v_add/sub u32, v_and/or/xor b32, v_mov
    slow  4.25     max/min/max3/min3/med3, shifts, bfe/bfi/perm/alignbit, every v_cvt, cndmask, v_cmp*, mbcnt, readlane/
                   writelane/readfirstlane, mad_u24, add3, lshl_add/lshl_or/and_or/or3, fma_mix, v_pk_*
    trans 8.2      rcp/rsq/sqrt/exp/log/sin/cos; v_mul_lo/hi_u32 priced here too (quarter rate)
usage: isa_census.py file.s <mangled-name-prefix> [loop-header-label | depth3]
Prints JSON: counts for the whole kernel and, when a loop header label (e.g. .LBB5_62) is given, for the lines from
that label to the last branch back to it (the queue-step loop of k_trace_q).
"""
import json
import re
import sys

FAST = ("v_fma_f32", "v_fmac_f32", "v_fmaak_f32", "v_fmamk_f32",    # the literal-constant forms: 2.5-2.6 cycles (profiles/r3_issue_rates3.jsonl)
         "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_mul_legacy_f32", "v_add_u32", "v_sub_u32",
        "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32",
        "v_not_b32", "v_mov_b64", "v_accvgpr")
TRANS = ("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32", "v_mad_u64_u32", "v_mad_i64_i32")
COST = {"fast": 2.6, "slow": 4.25, "trans": 8.24}


def classify(op):
    if not op.startswith("v_"):
        if op.startswith("s_"):
            return "salu" if not op.startswith(("s_load", "s_buffer_load", "s_waitcnt", "s_nop", "s_branch", "s_cbranch", "s_barrier", "s_endpgm")) else (
                "smem" if "load" in op else "ctrl")
        if op.startswith("ds_"):
            return "lds"
        if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            return "scratch" if op.startswith("scratch_") else "vmem"
        return "other"
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if base.startswith(TRANS):
        return "trans"
    if base.startswith(FAST):
        return "fast"
    return "slow"


def census(lines):
    cls, ops = {}, {}
    for ln in lines:
        m = re.match(r"\s+([a-z][a-z0-9_]+)\b", ln)
        if not m:
            continue
        op = m.group(1)
        c = classify(op)
        cls[c] = cls.get(c, 0) + 1
        if c in ("fast", "slow", "trans"):
            ops[op] = ops.get(op, 0) + 1
    valu = sum(cls.get(k, 0) for k in COST)
    mean = sum(cls.get(k, 0) * v for k, v in COST.items()) / max(1, valu)
    allops = dict(sorted(((re.sub(r"_(e32|e64|sdwa|dpp)$", "", o), 0) for o in ops), key=lambda kv: kv[0]))
    for o, n in ops.items():
        allops[re.sub(r"_(e32|e64|sdwa|dpp)$", "", o)] += n
    allops = dict(sorted(allops.items(), key=lambda kv: -kv[1]))
    return {"classes": cls, "valu": valu, "mean_issue_cycles_static": round(mean, 3),
            "slow_share_of_valu": round(cls.get("slow", 0) / max(1, valu), 3), "top_valu_opcodes_all": allops,
            "op_class": {o: classify(o) for o in allops}}


def main():
    path, name = sys.argv[1], sys.argv[2]
    header = sys.argv[3] if len(sys.argv) > 3 else None
    body, on = [], False
    for ln in open(path):
        if ln.startswith(name):
            on = True
        if on:
            body.append(ln)
            if ln.startswith(".Lfunc_end"):
                break
    out = {"kernel": name, "whole_kernel": census(body)}
    if header == "depth3":     # the largest loop below the outermost one (the queue-step loop of k_trace_q): its header label
        best = (0, None)
        for at, ln in enumerate(body):
            m = re.search(r"Loop Header: Depth=(\d+)", ln)
            if not m or int(m.group(1)) < 2:
                continue
            lab = next(body[i].split(":")[0] for i in range(at, -1, -1) if body[i].startswith(".LBB"))
            named = [i for i, l2 in enumerate(body) if ("Header=" + lab[2:] + " ") in l2]
            extent = (max(named) if named else at) - at
            if extent > best[0]:
                best = (extent, lab)
        header = best[1]
    if header:
        start = next(i for i, ln in enumerate(body) if ln.startswith(header + ":"))
        back = [i for i, ln in enumerate(body) if i > start and re.search(r"s_c?branch\w*\s+" + re.escape(header) + r"\b", ln)]
        # blocks that jump to the latch in front of the header belong to the loop too: take the last line that names the header
        named = [i for i, ln in enumerate(body) if i > start and ("Header=" + header[2:]) in ln]
        end = max(back + named) if back or named else len(body) - 1
        nxt = next((i for i in range(end + 1, len(body)) if body[i].startswith(".LBB")), len(body))
        out["loop"] = {"header": header, "lines": [start, nxt], **census(body[start:nxt])}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
