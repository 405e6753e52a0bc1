"""Round 6 (verdict r5 #1): a model of the path-tracing kernel's time that says where the SIMDs' idle cycles go -- and what hiding latency could recover -- BEFORE anything is built.

The kernel is a closed system: every SIMD holds N = 4 persistent wavefronts, and a wavefront runs scheduler iterations one after the other.  One iteration of one
wavefront needs (measured, per configuration):
    V    vector instructions -- they occupy the SIMD's issue port for c = 2.3 cycles each (profiles/r5_instruction_costs.txt: the kernels' own mix), but ONE
         wavefront cannot issue faster than one every 4 cycles (SQ_ACTIVE_INST_VALU counts 4 cycles per instruction: issuing_valu x wave cycles = 4 V, to the
         percent, on every configuration);
    M    a demand on the memory path that is SHARED (a CU's address unit and L1, the XCD's L2 and its fabric links): while one wavefront's gathers are served,
         another's wait -- a queueing station, visited by the 4 N wavefronts of a CU;
    Z    time that passes for the wavefront alone, whatever the others do: the unloaded latency of its ~9 dependent round trips, scalar instructions, LDS,
         dependency stalls -- a pure delay.
Mean-value analysis (Schweitzer's fixed point) of that network gives the iteration time T(N) = R_valu + R_mem + Z and the throughput N / T(N) per SIMD.
V and c are measured; M and Z are the two unknowns.  They are FITTED to two measurements -- the kernel's rate with 3 and with 4 wavefronts per SIMD (same pool per
wavefront: VR_BLOCKS_PER_CU) -- and the model is then CHECKED against five measurements it has not seen:
    the rate with 2 wavefronts per SIMD, with 256 and with 1024 idle cycles added to every pass of the hot pair (s_sleep: pure delay), with 64 dependency-free
    vector instructions added to every pass (V), and the fraction of its time a SIMD issues no vector instruction (counters: 1 - SQ_ACTIVE_INST_VALU x c / 4 per
    resident wavefront x N).
With the model in hand the candidates of the verdict are priced: what each could recover if it hid ALL of the latency it addresses, and at half of it.

usage: python tests/tools_latency_model.py [profiles/r6a_occupancy_and_padding.txt] > profiles/r6_latency_model.txt
Inputs: profiles/r6a_occupancy_and_padding.txt (tests/tools_r6_runs.sh call1 on one MI355X: occupancy and padding A/B of the round's kernels), profiles/r6_pmc_summary.json
(wave-cycle shares), profiles/r6_issue_budget.json (instructions and executions per iteration), profiles/r6_sched_stats.txt (cycles per event batch)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPI_WAVE = 4.0          # cycles one wavefront needs per vector instruction (issue cadence of a wave64 on a SIMD)


def mva(V, c, M, Z, n_simd, extra_delay=0.0, extra_valu=0.0, simds_per_mem=4, iters=4000):
    """Schweitzer approximate MVA.  Stations: the SIMD's vector issue port (demand (V + extra_valu) * c per iteration, n_simd customers; the part of the wavefront's own
    4-cycle cadence that the port does not see, (4 - c) per instruction, is delay), the memory path (demand M, simds_per_mem * n_simd customers), delay Z."""
    Vt = V + extra_valu
    Dv, Dm = Vt * c, M
    Zt = Z + extra_delay + Vt * (CPI_WAVE - c)
    K = simds_per_mem * n_simd
    Qv, Qm = n_simd / 3.0, K / 3.0
    for _ in range(iters):
        Rv = Dv * (1.0 + Qv * (n_simd - 1.0) / n_simd)
        Rm = Dm * (1.0 + Qm * (K - 1.0) / K)
        T = Rv + Rm + Zt
        Qv_n, Qm_n = n_simd * Rv / T, K * Rm / T
        if abs(Qv_n - Qv) + abs(Qm_n - Qm) < 1e-10:
            Qv, Qm = Qv_n, Qm_n
            break
        Qv, Qm = 0.5 * (Qv + Qv_n), 0.5 * (Qm + Qm_n)
    Rv = Dv * (1.0 + Qv * (n_simd - 1.0) / n_simd)
    Rm = Dm * (1.0 + Qm * (K - 1.0) / K)
    T = Rv + Rm + Zt
    X = n_simd / T                                   # iterations per cycle per SIMD
    return dict(T=T, X=X, util_valu=X * Dv, util_mem=X * simds_per_mem * Dm, Rv=Rv, Rm=Rm, Z=Zt)


def fit(V, c, x3_over_x4, T4):
    """M and Z such that the model gives the measured iteration time with 4 wavefronts per SIMD and the measured ratio of the rates with 3 and 4."""
    best = None
    lo, hi = 0.0, T4 / 4.0 / 4.0 * 1.05             # the memory path cannot be busier than 100 %: 16 M <= T4 (per CU) -> M <= T4 / 16
    for k in range(2001):
        M = lo + (hi - lo) * k / 2000.0
        # Z from T(4) = T4 by bisection (T is increasing in Z)
        a, b = 0.0, T4
        for _ in range(60):
            z = 0.5 * (a + b)
            if mva(V, c, M, z, 4)["T"] < T4:
                a = z
            else:
                b = z
        z = 0.5 * (a + b)
        r = (mva(V, c, M, z, 3)["X"]) / (mva(V, c, M, z, 4)["X"])
        err = abs(r - x3_over_x4)
        if best is None or err < best[0]:
            best = (err, M, z)
    return best[1], best[2], best[0]


def read_measurements(path):
    occ, ab = {}, {}
    for ln in open(path):
        m = re.match(r"== blocks_per_cu (\d) (\S+) \d+ \d+: kernel ms \S+ Msamples/s (\S+)", ln)
        if m:
            occ.setdefault(m.group(2), {})[int(m.group(1))] = float(m.group(3))
        m = re.match(r"== (\w+) (\S+) \d+ \d+: kernel ms \S+ Msamples/s (\S+)", ln)
        if m and not ln.startswith("== blocks_per_cu"):
            ab.setdefault(m.group(2), {}).setdefault(m.group(1), []).append(float(m.group(3)))
    return occ, {c: {v: sum(x) / len(x) for v, x in d.items()} for c, d in ab.items()}


def main():
    meas = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles/r6a_occupancy_and_padding.txt")
    occ, ab = read_measurements(meas)
    newest = lambda name: next(p for p in (os.path.join(ROOT, "profiles", r + name) for r in ("r6_", "r5_")) if os.path.exists(p))
    pmc = json.load(open(newest("pmc_summary.json")))
    bud = json.load(open(newest("issue_budget.json")))
    # cycles per execution of the event batches and lanes (instrumented kernels of this round: profiles/r6_sched_stats.txt), per configuration
    ev = {}
    cur = None
    p_stats = os.path.join(ROOT, "profiles/r6_sched_stats.txt")
    if os.path.exists(p_stats):
        for ln in open(p_stats):
            m = re.match(r"(\S+) \d+ \d+ \d+ thr", ln)
            if m:
                cur = m.group(1)
                ev[cur] = {}
            m = re.match(r"\s+(\w+)\s+exec\s+(\d+)\s+lanes/exec\s+(\S+).*cyc/exec\s+(\d+)", ln)
            if m and cur:
                ev[cur][m.group(1)] = (int(m.group(2)), float(m.group(3)), float(m.group(4)))
            m = re.match(r"\s+iterations/wave-sample\s+(\S+)", ln)
            if m and cur:
                ev[cur]["iters_per_wave_sample"] = float(m.group(1))
    print(__doc__.split("usage:")[0].rstrip())
    print()
    for name, key_occ, key_pmc, key_bud in (("c2", "c2", "c2", "c2"), ("c4 (512^3 dense, the north star's path)", "c4:512", "c4", "c4"), ("c5cloud", "c5cloud", "c5cloud", "c5cloud")):
        if key_occ not in occ:
            continue
        b, p = bud[key_bud], pmc[key_pmc]
        V = b["valu_per_sample_pmc"] / b["iterations_per_sample"]            # vector instructions per wavefront-iteration (hardware count)
        c = b["cycles_per_valu_op"]
        secs = b["sections"]
        passes = secs["march"]["executions_per_iteration"]                    # passes of the hot pair per iteration
        x4 = occ[key_occ][4]
        # cycles per iteration of ONE wavefront = 4 wavefronts per SIMD x SIMD cycles per sample / iterations per sample.  The absolute rate is the bench line's (whole
        # frames at the configuration's sample count: what the counters were taken on); the short runs of the occupancy / padding A/B (64-256 spp: more of a launch is
        # ramp and drain) only contribute RATIOS
        x_abs = 1024 * 2.4e3 / b["simd_cycles_per_sample"] if "simd_cycles_per_sample" in b else x4
        T4 = 4.0 * 1024 * 2.4e9 / (x_abs * 1e6 * b["iterations_per_sample"])
        M, Z, err = fit(V, c, occ[key_occ][3] / x4, T4)
        base = mva(V, c, M, Z, 4)
        sh = p["wave_cycles_share"]
        print("== %s" % name)
        print("measured: %.0f Msamples/s with 4 wavefronts per SIMD (bench line; %.0f in the A/B's short launches) = %.0f cycles per iteration of a wavefront (%.4f iterations per sample); V = %.0f vector instructions per iteration at c = %.2f cycles of the issue port each;" % (x_abs, x4, T4, b["iterations_per_sample"], V, c))
        print("          a wavefront's time by the counters (profiles/r6_pmc_summary.json): %.1f %% in vector instructions (4 V / T = %.1f %%), %.1f %% in other instructions, %.1f %% stalled at issue, %.1f %% waiting on memory" % (
            100 * sh["issuing_valu"], 100 * CPI_WAVE * V / T4, 100 * (sh["issuing"] - sh["issuing_valu"]), 100 * sh["issue_stalled"], 100 * sh["waiting_on_memory"]))
        print("fitted:   M = %.0f cycles of the shared memory path per iteration (its utilisation with 16 wavefronts per CU: %.0f %%), Z = %.0f cycles of private delay per iteration (of which %.0f are the wavefront's own 4-cycle issue cadence beyond the port's %.2f) -- residual of the fit %.4f" % (
            M, 100 * base["util_mem"], base["Z"], V * (CPI_WAVE - c), c, err))
        print("          at 4 wavefronts: issue port %.0f %% busy, iteration = %.0f at the issue port (queueing included) + %.0f at the memory path (queueing included) + %.0f private" % (100 * base["util_valu"], base["Rv"], base["Rm"], base["Z"]))
        rows = []
        rows.append(("3 wavefronts per SIMD (fitted)", occ[key_occ][3] / x4, mva(V, c, M, Z, 3)["X"] / base["X"]))
        rows.append(("2 wavefronts per SIMD", occ[key_occ][2] / x4, mva(V, c, M, Z, 2)["X"] / base["X"]))
        a = ab.get(key_occ, {})
        if "default" in a:
            d0 = a["default"]
            for var, label, kw in (("sleep4", "+256 idle cycles per hot-pair pass", dict(extra_delay=256.0 * passes)), ("sleep16", "+1024 idle cycles per hot-pair pass", dict(extra_delay=1024.0 * passes)),
                                   ("valu64", "+64 vector instructions per hot-pair pass", dict(extra_valu=64.0 * passes))):
                if var in a:
                    rows.append((label, a[var] / d0, mva(V, c, M, Z, 4, **kw)["X"] / base["X"]))
        slope = None
        if "default" in a and "sleep16" in a:
            slope = (1.0 - a["sleep16"] / a["default"]) / (1024.0 * passes)              # model-free: relative loss per cycle of private delay per iteration
        idle_meas = 1.0 - 4.0 * sh["issuing_valu"] * c / CPI_WAVE
        rows.append(("share of its time a SIMD issues no vector instruction", idle_meas, 1.0 - base["util_valu"]))
        print("check (not used by the fit, except the first line)                      measured    model")
        for label, m_, mo in rows:
            print("   %-68s %8.3f %8.3f   (%+.1f points)" % (label, m_, mo, 100 * (mo - m_)))
        # --- what the candidates could recover --------------------------------------------------------------------------------------------------------------
        wait = sh["waiting_on_memory"] * T4
        print("what-ifs (model): a wavefront waits on memory %.0f cycles per iteration; the memory path serves it for %.0f of them (queueing included), so ~%.0f are latency nobody is queueing for" % (wait, base["Rm"], max(0.0, wait - base["Rm"])))
        e = ev.get(key_occ) or ev.get(key_occ.split(":")[0])

        def per_iter(block, field):
            x = secs.get(block) or {}
            return x.get("executions_per_iteration", 0.0) * x.get(field, 0.0)
        hot_valu = per_iter("march", "valu_per_execution") + per_iter("collide", "valu_per_execution")
        ev_valu = sum(per_iter(k, "valu_per_execution") for k in ("nee", "postnee", "escape", "new"))
        # split of the latency between the event batches and the hot pair: the instrumented kernels' cycles per block (profiles/r6_sched_stats.txt), less the 4 cycles per
        # vector instruction the block executes itself
        share_events = None
        if e and all(k in e for k in ("nee", "postnee", "escape", "new", "march", "collide")):
            cyc = {k: e[k][0] * e[k][2] for k in ("nee", "postnee", "escape", "new", "march", "collide")}
            n_it = e["march"][0] / passes                                     # iterations of the instrumented launch
            ev_wait = max(0.0, sum(cyc[k] for k in ("nee", "postnee", "escape", "new")) / n_it - CPI_WAVE * ev_valu)
            hot_wait = max(0.0, (cyc["march"] + cyc["collide"]) / n_it - CPI_WAVE * hot_valu)
            share_events = ev_wait / (ev_wait + hot_wait)
            print("          of a wavefront's waiting, by the instrumented kernels' block times: %.0f %% inside the event batches (NEE's dependent warp levels, cold lines, texels), %.0f %% inside the hot pair" % (100 * share_events, 100 * (1 - share_events)))
        if share_events is None:
            share_events = 0.5
        lat = max(0.0, wait - base["Rm"])
        for label, hidden in (("(a) two path sets per lane in the hot pair, ALL of its latency hidden", lat * (1 - share_events)), ("(a) ... half of it (a realistic overlap)", 0.5 * lat * (1 - share_events)),
                              ("(b) event-batch phases interleaved with the hot pair, ALL of the batches' latency hidden", lat * share_events), ("(b) ... half of it", 0.5 * lat * share_events),
                              ("(a) + (b), all of it: every cycle of latency gone, queueing for the memory path and the issue port left", lat)):
            w = mva(V, c, M, Z, 4, extra_delay=-hidden)
            print("   %-100s -> x %.3f  (issue port %.0f %% busy, memory path %.0f %%)" % (label, w["X"] / base["X"], 100 * w["util_valu"], 100 * w["util_mem"]))
        if slope:
            print("   model-free cross-check: the 1024-cycle padding costs %.2f %% per 1000 cycles of delay per iteration -> all %.0f cycles of latency gone = x %.3f, the hot pair's share alone x %.3f" % (
                100 * slope * 1000, lat, 1.0 + slope * lat, 1.0 + slope * lat * (1 - share_events)))
        w5 = mva(V, c, M, Z, 5)
        print("   %-100s -> x %.3f  (profiles/r4e_*: measured x 0.97-0.99 at 96 registers and 120-slot pools)" % ("a fifth wavefront per SIMD with the same code and pool (not available: registers, LDS)", w5["X"] / base["X"]))
        print()


if __name__ == "__main__":
    main()
