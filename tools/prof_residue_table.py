"""Print the cycle split of the metric's kernels from a prof_residue.sh summary.json: every counter per launch, and -- where the counters
exist -- a wavefront's cycles as issuing / issue-stalled / parked (SQ_ACTIVE_INST_ANY + SQ_WAIT_INST_ANY + SQ_WAIT_ANY ~ SQ_WAVE_CYCLES,
MI355X_MICROARCH.md "rocprofv3 PMC slots"; all in quad-cycles) and the instruction-fetch figures.

    python3 tools/prof_residue_table.py <summary.json>
"""
import json
import sys


def main():
    s = json.load(open(sys.argv[1]))
    for name, e in s["configs"].items():
        print("== %s: %.3f ms per launch, %d units" % (name, e["ms_per_launch"], e["units"]))
        for kname, ke in e["kernels"].items():
            pc = ke.get("pmc_per_launch", {})
            if not pc:
                continue
            print("  -- %s  (%.3f ms)" % (kname, ke.get("ms_per_launch", float("nan"))))
            wc = pc.get("SQ_WAVE_CYCLES")
            for cn in sorted(pc):
                frac = ("   %.4f of SQ_WAVE_CYCLES" % (pc[cn] / wc)) if wc and cn.startswith(("SQ_WAIT", "SQ_ACTIVE", "SQ_INST_CYCLES", "SQ_BUSY")) else ""
                print("     %-30s %18.0f   %12.2f per unit%s" % (cn, pc[cn], pc[cn] / e["units"], frac))
            if wc and "SQ_ACTIVE_INST_ANY" in pc:
                rest = wc - pc.get("SQ_ACTIVE_INST_ANY", 0) - pc.get("SQ_WAIT_INST_ANY", 0) - pc.get("SQ_WAIT_ANY", 0)
                print("     split of wave cycles: issuing %.4f  issue-stalled %.4f  parked %.4f  unaccounted %.4f" % (
                    pc["SQ_ACTIVE_INST_ANY"] / wc, pc.get("SQ_WAIT_INST_ANY", 0) / wc, pc.get("SQ_WAIT_ANY", 0) / wc, rest / wc))
            if pc.get("SQC_ICACHE_REQ"):
                print("     icache: %.4f hit rate, %.1f requests per unit" % (pc.get("SQC_ICACHE_HITS", 0) / pc["SQC_ICACHE_REQ"], pc["SQC_ICACHE_REQ"] / e["units"]))


if __name__ == "__main__":
    main()
