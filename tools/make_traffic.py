#!/usr/bin/env python3
"""tools/make_traffic.py COMMIT TAG...  : profiles/traffic.json entries from the CSVs of tools/collect_r03_pmc.sh
(gpurun_out/pmc_r03_<TAG>/pass*/), and a text summary per tag under profiles/r03_bench_<TAG>_pmc.txt.

HBM bytes per launch of the config's kernel(s): read = TCC_EA0_RDREQ_128B*128 + RDREQ_64B*64 + RDREQ_32B*32 (the by-size
counters, which is what MI355X_MICROARCH.md's gfx950 correction of FETCH_SIZE amounts to), write = TCC_EA0_WRREQ_64B*64 + the
rest of WRREQ * 32.  VALU: instructions per output pixel-lane (SQ_INSTS_VALU * 64 / pixels) and the share of all SIMD cycles with a
VALU instruction issuing (SQ_ACTIVE_INST_VALU counts quad-cycles summed over the SIMDs; GRBM_GUI_ACTIVE is summed over the 8
XCDs), averaged over the dispatches of the run."""
import collections, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

KERNELS = {"c4": ["k_undistort_rotate_rp", "k_superview_cubic"], "c4_fused": ["k_undistort_rotate_superview"],
           "c5": ["k_undistort_rotate_blend"]}
N_SIMD, N_XCD = 1024, 8


def counters(tag, kern):
    vals = {}
    for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_r03_" + tag, "pass*", "**", "*counter_collection.csv"), recursive=True)):
        acc = collections.defaultdict(list)
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if kern in row["Kernel_Name"] and "scratch_put" not in row["Kernel_Name"]:
                    acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            vals[k] = (sum(v) / len(v), len(v))
    return vals


def main():
    commit = sys.argv[1]
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    with open(tpath) as f:
        table = json.load(f)
    for tag in sys.argv[2:]:
        cfg, _, b = tag.partition("_b") if tag.rsplit("_b", 1)[-1].isdigit() else (tag, "", "")
        nb = bench.BLEND.get(cfg, 1)
        B = int(b) if b else (64 if cfg.startswith("c5") else 128)
        B -= B % nb
        _, iw, ih, ow, oh, dt, desc, _ = bench.CONFIGS[cfg]
        lines, rd, wr, insts, active, gui = [], 0.0, 0.0, 0.0, 0.0, 0.0
        for kern in KERNELS.get(cfg, ["k_undistort_rotate"]):
            v = counters(tag, kern)
            if not v:
                print("no counters for", tag, kern); sys.exit(1)
            g = lambda k: v.get(k, (0.0, 0))[0]
            r = g("TCC_EA0_RDREQ_128B_sum") * 128 + g("TCC_EA0_RDREQ_64B_sum") * 64 + g("TCC_EA0_RDREQ_32B_sum") * 32
            w = g("TCC_EA0_WRREQ_64B_sum") * 64 + (g("TCC_EA0_WRREQ_sum") - g("TCC_EA0_WRREQ_64B_sum")) * 32
            rd += r; wr += w
            insts += g("SQ_INSTS_VALU"); active += g("SQ_ACTIVE_INST_VALU"); gui += g("GRBM_GUI_ACTIVE")
            lines.append("== %s (%s)" % (kern, tag))
            for k, (val, n) in v.items():
                lines.append("%-34s %18.1f   (avg of %d dispatches)" % (k, val, n))
            lines.append("read %.0f B + write %.0f B per launch; VALU busy %.3f of SIMD cycles" % (r, w, g("SQ_ACTIVE_INST_VALU") * 4 / (N_SIMD * g("GRBM_GUI_ACTIVE") / N_XCD)))
        # pixel-lanes the VALU count is spread over: every input frame's output-size pixels (the blend kernel remaps each input frame)
        px = B * ow * oh + (B * bench.STRETCH[cfg][0] * bench.STRETCH[cfg][1] if cfg in bench.STRETCH else 0)
        out = os.path.join(ROOT, "profiles", "r03_bench_%s_pmc.txt" % tag)
        with open(out, "w") as f:
            f.write("\n".join(lines) + "\n")
        table["%s_b%d" % (cfg, B)] = {
            "hbm_bytes_per_launch": int(rd + wr), "read_bytes": int(rd), "write_bytes": int(wr), "commit": commit,
            "valu": {"insts_per_pixel": insts * 64 / px, "frac": active * 4 / (N_SIMD * gui / N_XCD),
                     "what": "SQ_INSTS_VALU*64 / pixels computed per launch (C4: intermediate + output pixels); frac = SQ_ACTIVE_INST_VALU*4 / (1024 SIMDs * GRBM_GUI_ACTIVE/8): share of all SIMD cycles with a VALU instruction issuing"},
            "source": "profiles/%s: `python3 bench.py --config %s%s --cpu-frames 0 --no-pcie --steps 3 --warmup 1` under tools/collect_r03_pmc.sh (%s)" % (
                os.path.basename(out), cfg, " --batch %s" % b if b else "", desc)}
        print(tag, "->", json.dumps(table["%s_b%d" % (cfg, B)]["valu"]["frac"]), int(rd + wr))
    with open(tpath, "w") as f:
        json.dump(table, f, indent=1)


if __name__ == "__main__":
    main()
