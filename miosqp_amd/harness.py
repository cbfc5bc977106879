"""Statistics of the reference's benchmark scripts, restated for the harness in examples/.

  switching_frequency   /root/reference/examples/power_converter/power_converter.py:549-569 with
                        utils.compute_on_transitions (utils.py:80-97)
  thd                   utils.get_thd / get_phase_thd / get_dft (utils.py:100-215)
  timing_row            the columns of results/power_converter_timings.csv (run_example.py:92-130)

No plant model here: the closed-loop signals come from a recorded run of the reference (tests/golden/
make_power_converter_long.py); what the harness re-runs is the MIQP sequence, on the engine under test.
"""
import numpy as np


def on_transitions(u, u_prev):
    """ON transitions of the 12 semiconductor switches between two three-level inputs in {-1, 0, 1}^3: per phase,
    switch 0 turns on going 0 -> 1, switch 2 going 1 -> 0, switch 3 going 0 -> -1, switch 1 going -1 -> 0."""
    out = np.zeros(12)
    for ph in range(3):
        a, b = u_prev[ph], u[ph]
        if a == 0 and b == 1:
            out[4 * ph + 0] = 1
        elif a == 1 and b == 0:
            out[4 * ph + 2] = 1
        elif a == 0 and b == -1:
            out[4 * ph + 3] = 1
        elif a == -1 and b == 0:
            out[4 * ph + 1] = 1
    return out


def switching_frequency(U, init_periods, sim_periods, nstpp, freq):
    """Average device switching frequency [Hz] over the measured periods: ON transitions per switch divided by the
    measured time, averaged over the 12 switches."""
    n_sw = np.zeros(12)
    for i in range(int(init_periods * nstpp), U.shape[1]):
        n_sw += on_transitions(U[:3, i], U[:3, i - 1])
    return float(np.mean(n_sw / (1.0 / freq * sim_periods)))


def _dft(signal, time, freq):
    """One-sided amplitude spectrum over a whole number of fundamental periods (so the lines are sharp), scaled to be
    independent of the window length, padded to the one-sided length of the full signal."""
    ts = np.mean(np.diff(time))
    n_samples = (1.0 / freq) / ts
    n_period = np.floor(len(signal) / n_samples)
    if n_period <= 0:
        raise ValueError("DFT: signal is too short; less than one fundamental period!")
    x = signal[:int(n_period * n_samples)]
    n, ns = len(x), len(signal)
    m = np.fft.fft(x) / n
    m = m[:(n + 1) // 2] if n % 2 else m[:n // 2 + 1]
    m[1:] = m[1:] * 2.0
    nf = (ns + 1) // 2 + 1 if ns % 2 else ns // 2 + 1
    return np.append(m, np.zeros(nf - len(m)))


def thd(Y, time, freq):
    """Total harmonic distortion [%] of a three-phase quantity (samples x 3, per-unit peak): per phase the
    fundamental line and its two neighbours are removed, THD = 100 sqrt(sum |ripple|^2); mean over the phases."""
    if Y.shape[1] != 3:
        raise ValueError("Y must be samples x 3")
    out = np.zeros(3)
    for ph in range(3):
        m = _dft(Y[:, ph], time, freq)
        i_fund = int(np.argmax(np.abs(m)))
        m[max(0, i_fund - 1):i_fund + 2] = 0.0
        out[ph] = 100.0 * np.sqrt(np.sum(np.abs(m) ** 2))
    return float(np.mean(out))


def closed_loop_statistics(pc):
    """fsw [Hz] and THD [%] of a recorded closed-loop run, as Model.get_statistics computes them."""
    t0 = int(pc["init_periods"] * pc["Nstpp"])
    fsw = switching_frequency(pc["U"], pc["init_periods"], pc["sim_periods"], pc["Nstpp"], pc["freq"])
    return fsw, thd(pc["Y_phase"][:, t0:].T, pc["t"][t0 + 1:], pc["freq"])


def timing_row(horizon, records, first_timed):
    """One row of power_converter_timings.csv: solve-time statistics over the measured steps (seconds, as the
    reference keeps them), the relaxation solver's share of the run time in percent and the iterations per node,
    both averaged over ALL steps (simulate_cl, power_converter.py:640-665)."""
    t = np.array([r["run_time"] for r in records[first_timed:]])
    share = np.mean([100.0 * r["osqp_solve_time"] / r["run_time"] for r in records])
    return dict(T=horizon, miosqp_avg=float(np.mean(t)), miosqp_std=float(np.std(t)), miosqp_min=float(np.min(t)),
                miosqp_max=float(np.max(t)), miosqp_osqp_avg_time=float(share),
                miosqp_avg_osqp_iter=float(np.mean([r["osqp_iter_avg"] for r in records])))
