#!/usr/bin/env python3
"""Build profiles/<round>_k1_valu_model.json (round = VKR_ROUND, default r3) (what bench.py's `roofline` objects read) from the evidence files of
tools/profile_round3.sh.  usage: make_valu_model.py <tag>:<workload> [<tag>:<workload> ...]
reads  profiles/<tag>_pmc_summary.txt, profiles/<tag>_bench.json, profiles/<round>_k1_isa_census.json,
       profiles/r2_issue_rates.jsonl + r2_issue_rates2.jsonl (calibrated issue cycles per opcode)

VALU issue bound of k_trace_q.  The SQ counters give the dynamic number of VALU wave-instructions and a split into
ADD/MUL/FMA_F32 (fast class), TRANS_F32, CVT (slow class), INT32 and the rest (moves, compares, selects, min/max, lane ops).
INT32 and the rest are not resolved into calibrated classes by any counter, so two prices are stated:
  mean_issue_cycles_lower     unresolved instructions priced at the FAST class  -> peak is a true upper bound, frac a lower
                              bound of the VALU-issue utilisation (this is what bench.py reports as peak / frac)
  mean_issue_cycles_estimate  unresolved instructions priced at the static class mix of the queue-step loop's unresolved
                              opcodes (tools/isa_census.py)                     -> the working estimate (frac_estimate)
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
ROUND = os.environ.get("VKR_ROUND", "r3")
sys.path.insert(0, ROOT)
from vulkanrenderer_amd.fingerprint import kernel_source_fingerprint   # noqa: E402
FAST, SLOW, TRANS = 2.6, 4.25, 8.24          # cycles per wave-instruction per SIMD (r2_issue_rates*.jsonl, class means)


def summary(path):
    out, k = {}, None
    for ln in open(path):
        if not ln.startswith(" "):
            k = ln.strip()
            out[k] = {}
        else:
            m = re.match(r"\s+(\S+)\s+mean/dispatch\s+(\S+)", ln)
            out[k][m.group(1)] = float(m.group(2))
    return out


def main():
    census = json.load(open(os.path.join(P, f"{ROUND}_k1_isa_census.json")))
    # static class mix of the loop's opcodes that no SQ_INSTS_VALU_* class counter resolves
    ops = census["loop"]["top_valu_opcodes_all"]
    resolved = re.compile(r"v_(add|sub|subrev|mul|fma|fmac|fmaak|fmamk|mul_legacy|pk_add|pk_mul|pk_fma)_f32|v_cvt|v_(rcp|rsq|sqrt|exp|log|sin|cos)")
    un = {o: n for o, n in ops.items() if not resolved.match(o)}
    un_cost = sum(n * {"fast": FAST, "slow": SLOW, "trans": TRANS}[census["loop"]["op_class"][o]] for o, n in un.items()) / sum(un.values())
    # K2 (k_atrous_tile) has no inner loop worth isolating: the whole kernel's static mix prices its unresolved instructions
    census2 = json.load(open(os.path.join(P, f"{ROUND}_k2_isa_census.json")))["whole_kernel"]
    un2 = {o: n for o, n in census2["top_valu_opcodes_all"].items() if not resolved.match(o)}
    un2_cost = sum(n * {"fast": FAST, "slow": SLOW, "trans": TRANS}[census2["op_class"][o]] for o, n in un2.items()) / sum(un2.values())
    model = {"_doc": __doc__.strip().split("\n\n")[1].replace("\n", " "),
             "class_cycles": {"fast": FAST, "slow": SLOW, "trans": TRANS, "source": "profiles/r2_issue_rates.jsonl, profiles/r2_issue_rates2.jsonl, profiles/r3_issue_rates3.jsonl"},
             "unresolved_opcodes_static_cost": round(un_cost, 3), "census": f"profiles/{ROUND}_k1_isa_census.json",
             "source_fingerprint": kernel_source_fingerprint(),
             "source_fingerprint_note": "sha256 over the kernel sources these counters were measured on (vulkanrenderer_amd/fingerprint.py); bench.py and "
                                        "tests/test_host_logic.py compare it with the sources in the tree, so a kernel change without a re-profile is reported, not replayed"}
    for arg in sys.argv[1:]:
        tag, wl = arg.split(":")
        s = summary(os.path.join(P, f"{tag}_pmc_summary.txt"))
        bench = json.loads(open(os.path.join(P, f"{tag}_bench.json")).read().strip().splitlines()[-1])
        k1 = next(v for k, v in s.items() if "k_trace_q<false" in k)
        k2 = next(v for k, v in s.items() if "k_atrous_tile<false" in k)
        n = k1["SQ_INSTS_VALU"]
        fast = k1["SQ_INSTS_VALU_ADD_F32"] + k1["SQ_INSTS_VALU_MUL_F32"] + k1["SQ_INSTS_VALU_FMA_F32"]
        trans, cvt, int32 = k1["SQ_INSTS_VALU_TRANS_F32"], k1["SQ_INSTS_VALU_CVT"], k1["SQ_INSTS_VALU_INT32"]
        rest = n - fast - trans - cvt - int32
        lower = (fast * FAST + trans * TRANS + cvt * SLOW + (int32 + rest) * FAST) / n
        est = (fast * FAST + trans * TRANS + cvt * SLOW + (int32 + rest) * un_cost) / n
        cu_cycles = k1["GRBM_GUI_ACTIVE"] / 8.0 * 256.0          # GRBM_GUI_ACTIVE sums the 8 XCDs; TA/TD/TCP *_sum sum the 256 CUs
        ao_rays = bench["kernels"]["ao_mrays_per_s_kernel"] * 1e3 * bench["kernels"]["trace_ms"]      # AO + retry rays of one K1 launch
        iters = 4
        wh = re.search(r" (\d+)x(\d+),", bench["config"]["workload"])          # AO resolution of the profiled run
        k2n = k2["SQ_INSTS_VALU"]
        k2_fast = k2["SQ_INSTS_VALU_ADD_F32"] + k2["SQ_INSTS_VALU_MUL_F32"] + k2["SQ_INSTS_VALU_FMA_F32"]
        k2_un = k2n - k2_fast - k2["SQ_INSTS_VALU_TRANS_F32"] - k2["SQ_INSTS_VALU_CVT"]
        k2_lower = (k2_fast * FAST + k2["SQ_INSTS_VALU_TRANS_F32"] * TRANS + k2["SQ_INSTS_VALU_CVT"] * SLOW + k2_un * FAST) / k2n
        k2_est = (k2_fast * FAST + k2["SQ_INSTS_VALU_TRANS_F32"] * TRANS + k2["SQ_INSTS_VALU_CVT"] * SLOW + k2_un * un2_cost) / k2n
        model[wl] = {
            "kernel": next(k for k in s if "k_trace_q<false" in k),
            "source": f"profiles/{tag}_pmc_summary.txt (rocprofv3 --pmc, one counter group per run, mean over the dispatches of a 4-frame bench run); "
                      f"rays from profiles/{tag}_bench.json",
            "valu_insts_per_launch": n, "salu_insts_per_launch": k1["SQ_INSTS_SALU"], "lds_insts_per_launch": k1["SQ_INSTS_LDS"],
            "vmem_rd_insts_per_launch": k1["SQ_INSTS_VMEM_RD"],
            "rays_per_launch": ao_rays, "valu_insts_per_ray": n / ao_rays,
            "valu_mix": {"f32_add_mul_fma (fast)": fast / n, "trans": trans / n, "cvt (slow)": cvt / n, "int32 (unresolved)": int32 / n,
                         "rest: mov/cmp/cndmask/min/max/lane ops (unresolved)": rest / n},
            "mean_issue_cycles_lower": round(lower, 3), "mean_issue_cycles_estimate": round(est, 3),
            "mean_issue_cycles_per_valu_inst": round(lower, 3),
            "fetch_size_kb": k1["FETCH_SIZE"], "write_size_kb": k1["WRITE_SIZE"],
            "hbm_bytes_per_launch": k1["FETCH_SIZE"] * 1024.0 * 2.0 + k1["WRITE_SIZE"] * 1024.0,
            "hbm_correction": "MI355X_MICROARCH.md HBM section: unit KB (x1024); FETCH_SIZE doubled (gfx950 reports half the bytes of 16-B-per-lane reads; "
                              "an upper estimate for these gathers), WRITE_SIZE as is; fabric-side counters, Infinity-Cache hits included",
            "td_busy_frac": k1["TD_TD_BUSY_sum"] / cu_cycles, "ta_busy_frac": k1["TA_TA_BUSY_sum"] / cu_cycles,
            "td_tc_stall_frac": k1["TD_TC_STALL_sum"] / cu_cycles, "tcp_pending_stall_frac": k1["TCP_PENDING_STALL_CYCLES_sum"] / cu_cycles,
            "l1_hit_rate": 1.0 - k1["TCP_TCC_READ_REQ_sum"] / k1["TCP_TOTAL_CACHE_ACCESSES_sum"],
            "l2_hit_rate": k1["TCC_HIT_sum"] / (k1["TCC_HIT_sum"] + k1["TCC_MISS_sum"]),
            "wave_cycles_issue_stalled_frac": k1["SQ_WAIT_INST_ANY"] / k1["SQ_WAVE_CYCLES"],
            "k2_hbm_bytes_per_iteration": k2["FETCH_SIZE"] * 1024.0 + k2["WRITE_SIZE"] * 1024.0,
            "k2_hbm_note": "k_atrous_tile: FETCH_SIZE x1024 uncorrected (the kernel's own read volume: (8 B normal + 2 B depth + 2 B ao) x (36x12)/(32x8) "
                           "halo factor 1.69 = 20 B/px) + WRITE_SIZE x1024",
            "k2_valu_insts_per_iteration": k2["SQ_INSTS_VALU"], "k2_iterations": iters,
            "k2_pixels": int(wh.group(1)) * int(wh.group(2)),
            "k2_mean_issue_cycles_lower": round(k2_lower, 3), "k2_mean_issue_cycles_estimate": round(k2_est, 3),
            "k2_valu_mix": {"f32_add_mul_fma (fast)": k2_fast / k2n, "trans": k2["SQ_INSTS_VALU_TRANS_F32"] / k2n, "cvt (slow)": k2["SQ_INSTS_VALU_CVT"] / k2n,
                            "int32 + rest (unresolved)": k2_un / k2n},
            "k2_census": f"profiles/{ROUND}_k2_isa_census.json", "k2_unresolved_opcodes_static_cost": round(un2_cost, 3),
        }
    json.dump(model, open(os.path.join(P, f"{ROUND}_k1_valu_model.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in model.items() if k in ("c3", "c4")}, indent=1))


if __name__ == "__main__":
    main()
