"""rocprofv3 PMC passes -> MFMA utilisation and HBM bandwidth per kernel (profiles/r03_mfma_util_hbm_bs64_bf16.{txt,json}).

  python tools/pmc_util.py <sq_pass.csv> <fetch_pass.csv> <write_pass.csv> <plain_kernel_trace.csv> <calibration_sq_pass.csv> <out prefix> [what was profiled]

* MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x GRBM_GUI_ACTIVE): the counter adds up the cycles the matrix pipe of every SIMD is
  busy (= 32 x the number of v_mfma_f32_32x32x16_bf16, 16 x the 16x16x32 form -- checked against SQ_INSTS_MFMA), GRBM_GUI_ACTIVE the
  shader-clock cycles of the dispatch summed over the 8 XCD instances: busy / active = 1024 SIMDs / 8 = 128 at 100 %.  The normalisation is
  CHECKED on a pure-MFMA loop (tools/ubench/mfma_rate.py mode 0: back-to-back independent MFMAs from registers, two waves per SIMD), which
  must read close to 1 (measured 0.92: 33.5 instead of 32 cycles per MFMA by s_memtime, plus the ramp of a 0.5 ms launch).
* HBM GB/s = (2 x FETCH_SIZE + WRITE_SIZE) KB per launch (MI355X_MICROARCH.md 'HBM': gfx950 tallies 128-byte requests at 64 B) / the launch's
  duration in the PLAIN kernel trace (the PMC passes serialise and slow the launches down).
* frac_of_peak (bench.py's roofline.frac) prices the algorithmic FLOPs against 2.5 PFLOP/s = 2.4 GHz; mfma_util counts cycles at the clock
  the chip actually ran (1.6 - 2.1 GHz under this load, tools/ubench/mfma_rate.py): util = frac x 2.4 GHz / clock x (padded / algorithmic FLOPs)."""
import csv
import json
import os
import re
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("unsigned short", "bf16").replace("void ", "")
    name = re.sub(r"\(.*$", "", name)
    return name.replace(", ", ",").strip()


def counters(path):
    tot = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
    return tot, cnt


def durations(path, marks="pack_image_kernel", last=4):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    m = [i for i, r in enumerate(rows) if marks in r["Kernel_Name"]]
    seg = rows[m[-last - 1]:m[-1]] if len(m) > last else rows
    d, c = defaultdict(float), defaultdict(int)
    for r in seg:
        k = short(r["Kernel_Name"])
        d[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
        c[k] += 1
    n = last if len(m) > last else 1
    return {k: (d[k] / c[k], c[k] / n) for k in d}


def main():
    sq, sqn = counters(sys.argv[1])
    ft, fn = counters(sys.argv[2])
    wt, wn = counters(sys.argv[3])
    dur = durations(sys.argv[4])
    cal, _ = counters(sys.argv[5])
    prefix = sys.argv[6]
    ck = [k for k in cal if k.startswith("k<0>")]
    assert ck, list(cal)
    c0 = cal[ck[0]]
    norm = 128.0   # 1024 SIMDs / 8 XCD instances of GRBM_GUI_ACTIVE
    out = {"__calibration__": {"kernel": "tools/ubench/mfma_rate.hip k<0> (pure MFMA loop, 2 waves/SIMD)",
                               "mfma_util_of_the_pure_mfma_loop": c0["SQ_VALU_MFMA_BUSY_CYCLES"] / c0["GRBM_GUI_ACTIVE"] / norm,
                               "mfma_busy_cycles_per_instruction": c0["SQ_VALU_MFMA_BUSY_CYCLES"] / max(c0.get("SQ_INSTS_MFMA", 0.0), 1.0)}}
    rows = []
    for k, (t, per_step) in dur.items():
        e = {"launches_per_step": per_step, "avg_us": t * 1e6}
        if k in sq and sq[k].get("GRBM_GUI_ACTIVE", 0) > 0:
            e["mfma_util"] = sq[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / sq[k]["GRBM_GUI_ACTIVE"] / norm
        f = ft.get(k, {}).get("FETCH_SIZE", 0.0) * 1024 * 2 / max(fn.get(k, {}).get("FETCH_SIZE", 1), 1)
        w = wt.get(k, {}).get("WRITE_SIZE", 0.0) * 1024 / max(wn.get(k, {}).get("WRITE_SIZE", 1), 1)
        e.update(fetch_bytes_per_launch=f, write_bytes_per_launch=w, traffic_bytes_per_launch=f + w, hbm_gbps=(f + w) / t / 1e9)
        out[k] = e
        rows.append((t * per_step, k, e))
    import bench

    out["__kernel_source_sha256_16__"] = bench.kernel_source_hash()
    json.dump(out, open(prefix + ".json", "w"), indent=1, sort_keys=True)
    rows.sort(reverse=True)
    with open(prefix + ".txt", "w") as f:
        f.write("# MFMA utilisation and HBM bandwidth per kernel of the %s (tools/pmc_util.py; rocprofv3 --pmc passes:\n" % (sys.argv[7] if len(sys.argv) > 7 else "bs=64 bf16 training step"))
        f.write("#   SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE | FETCH_SIZE | WRITE_SIZE, each with --kernel-trace; durations from\n")
        f.write("#   a plain --kernel-trace run, last 4 steps).  mfma_util = MFMA-pipe busy cycles / (SIMDs x active cycles), normalised on a pure-MFMA loop\n")
        f.write("#   = busy / active / 128 (check: the pure-MFMA loop reads %.3f, %.1f busy cycles per MFMA instruction); HBM GB/s = (2 x FETCH_SIZE + WRITE_SIZE) / duration.\n" % (
            out["__calibration__"]["mfma_util_of_the_pure_mfma_loop"], out["__calibration__"]["mfma_busy_cycles_per_instruction"]))
        f.write("%-58s %8s %9s %10s %10s %10s %9s\n" % ("kernel", "n/step", "avg us", "mfma_util", "fetch MB", "write MB", "HBM GB/s"))
        for _, k, e in rows[:24]:
            f.write("%-58s %8.1f %9.1f %10s %10.1f %10.1f %9.0f\n" % (k[:58], e["launches_per_step"], e["avg_us"],
                                                                  ("%.3f" % e["mfma_util"]) if "mfma_util" in e else "-",
                                                                  e["fetch_bytes_per_launch"] / 1e6, e["write_bytes_per_launch"] / 1e6, e["hbm_gbps"]))
    print(open(prefix + ".txt").read())


if __name__ == "__main__":
    main()
