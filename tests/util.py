"""Shared helpers of the test-suite: golden cases, input regeneration, tolerances."""
import hashlib
import os

import numpy as np

import pyoracle as po
from syllable_detector_swift_amd import nets, synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# Floating-point tolerance of the north star ("spectrogram/NN activations within 1e-5 fp32"):
# absolute 1e-5 at unit scale, relative 1e-5 against the largest value of the same spectrogram
# column / output vector (rounding error of an FFT bin scales with the column, not the bin).
TOL = 1e-5


def sample_net():
    return nets.from_npz(os.path.join(GOLD, "sample_net.npz"))


def template():
    return np.load(os.path.join(GOLD, "syllable_template.npy"))


def case_names():
    return sorted(f[:-4] for f in os.listdir(GOLD) if f.startswith("case_") and f.endswith(".npz"))


def regenerate(kind: str, seed: int, n: int) -> np.ndarray:
    if kind == "channel":
        return synth.channel(n, seed)
    if kind == "channel_fs16000":
        return synth.channel(n, seed, fs=16000.0)
    if kind == "syllable_channel":
        return synth.syllable_channel(n, template(), seed=seed)
    if kind == "syllable_channel_hop128":
        return synth.syllable_channel(n, template(), seed=seed, hop=128)
    raise KeyError(kind)


def load_case(name: str):
    """-> (cfg, samples, golden dict).  Inputs not stored explicitly are regenerated from their
    seed and checked against the stored sha256."""
    z = np.load(os.path.join(GOLD, name + ".npz"))
    tmp = {k[4:]: z[k] for k in z.files if k.startswith("cfg_")}
    path = os.path.join(GOLD, "_tmp_%s_%d.npz" % (name, os.getpid()))
    np.savez(path, **tmp)
    try:
        cfg = nets.from_npz(path)
    finally:
        os.remove(path)
    cfg.window = int(z["window"][0])
    cfg.spectrum = int(z["spectrum"][0])
    cfg.rule = int(z["rule"][0])
    n = int(z["n_samples"][0])
    if "samples" in z.files:
        x = z["samples"]
    else:
        x = regenerate(str(z["desc_kind"]), int(z["desc_seed"]), n)
    digest = hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest()
    assert digest == str(z["sha256"]), "regenerated input of %s differs from the one the golden was made from" % name
    gold = {k: z[k] for k in z.files if not k.startswith("cfg_") and k != "samples"}
    gold["flags"] = np.unpackbits(z["flags"])[: int(z["n_evals"][0])]
    return cfg, x, gold


def oracle_for(cfg) -> po.Oracle:
    return po.Oracle(po.from_config(cfg))


def column_scale(cols: np.ndarray) -> np.ndarray:
    """Per-column tolerance scale: max(1, max |column|)."""
    return np.maximum(1.0, np.abs(cols).max(axis=-1, keepdims=True))


def assert_columns_close(got, want, tol=TOL):
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == want.shape
    err = np.abs(got - want) / column_scale(want)
    assert err.max() <= tol, "spectrogram error %.3g (scaled) > %.1g" % (err.max(), tol)


def assert_outputs_close(got, want, tol=TOL):
    """`tol`: one bar, or one per evaluation (first axis)."""
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == want.shape
    err = np.abs(got - want) / np.maximum(1.0, np.abs(want))
    bar = np.asarray(tol, np.float64).reshape((-1,) + (1,) * (err.ndim - 1))
    worst = np.unravel_index(np.argmax(err / bar), err.shape)
    assert (err <= bar).all(), "output error %.3g > %.1g at %s" % (err[worst], np.broadcast_to(bar, err.shape)[worst], worst)


def assert_flags_exact(got_flags, want_out64, thresholds, rule, tol=TOL):
    """Flags must be bit-identical wherever the anchor's output is farther than the float
    tolerance from its threshold; the committed goldens keep every evaluation farther."""
    thr = np.asarray(thresholds, np.float64)[None, :]
    o = np.asarray(want_out64, np.float64)
    hit = o >= thr
    want = hit[:, 0] if rule == 0 else hit.any(axis=1)
    cols = slice(0, 1) if rule == 0 else slice(None)
    bar = np.asarray(tol, np.float64).reshape(-1, 1)                        # one bar, or one per evaluation
    safe = (np.abs(o - thr)[:, cols] > 2 * bar * np.maximum(1.0, np.abs(o[:, cols]))).all(axis=1)
    got = np.asarray(got_flags).astype(bool)
    assert got.shape == want.shape
    bad = np.nonzero((got != want) & safe)[0]
    assert bad.size == 0, "flag mismatch at evaluations %s" % bad[:8]
    return safe


def band_condition(o: "po.Oracle", cfg, x: np.ndarray, cols64: np.ndarray = None) -> np.ndarray:
    """Per evaluation: (norm of the window's whole one-sided spectra) / (norm of its in-band columns).  Any fp32 transform
    leaves an error of about 2^-24 of a frame's whole energy in every bin, so a band that holds 1/kappa of the norm is known
    to about kappa 2^-23 relative -- whatever the evaluation order (the reference's vDSP FFT included).  l2normalize turns
    that into an absolute error of the network's input; tests scale the bar of such evaluations with kappa."""
    W, N, T = cfg.windowLength, cfg.fourierLength, cfg.timeRange
    gap = max(0, -cfg.windowOverlap)
    hop = gap + W - max(0, cfg.windowOverlap)
    if cols64 is None:
        cols64 = o.spectrogram(x, po.F64)
    J = cols64.shape[0]
    w = o.window().astype(np.float64)
    fr = np.lib.stride_tricks.sliding_window_view(x.astype(np.float64)[gap:], W)[::hop][:J]
    full = 0.5 * N * ((fr * w[None, :]) ** 2).sum(axis=1)            # Parseval: sum over the one-sided bins of |X|^2
    band = (cols64 ** 2).sum(axis=1)
    E = J - T + 1
    cf, cb = np.concatenate([[0.0], np.cumsum(full)]), np.concatenate([[0.0], np.cumsum(band)])
    num, den = cf[T:T + E] - cf[:E], cb[T:T + E] - cb[:E]
    with np.errstate(divide="ignore", invalid="ignore"):
        k = np.sqrt(num / den)
    return np.where(np.isfinite(k), k, 1.0)


def log_condition(o: "po.Oracle", cfg, x: np.ndarray, cols64: np.ndarray, evals, trials: int = 12) -> np.ndarray:
    """For log / dB columns (and, with the identity for the logarithm, any other chain): how far the anchor's own output moves when every bin's amplitude moves by +-2^-23 of the norm of
    its frame's whole spectrum -- the error any fp32 transform leaves in a bin (band_condition's premise; a bin 60 dB under
    the rest of its frame is known to 1e-4 of itself, and the logarithm turns that into an absolute error of the network's
    input) and, on top, every scaled value moves by half an fp32 ulp of itself (the reference's own columns are floats).  Per evaluation of `evals`: the largest move over a few random sign patterns, fp64 network on perturbed columns."""
    W, N, T = cfg.windowLength, cfg.fourierLength, cfg.timeRange
    gap = max(0, -cfg.windowOverlap)
    hop = gap + W - max(0, cfg.windowOverlap)
    J = cols64.shape[0]
    w = o.window().astype(np.float64)
    fr = np.lib.stride_tricks.sliding_window_view(x.astype(np.float64)[gap:], W)[::hop][:J]
    delta = 2.0 ** -23 * np.sqrt(0.5 * N * ((fr * w[None, :]) ** 2).sum(axis=1))      # amplitude units of the columns
    power = cols64.min() >= 0.0 and getattr(cfg, "spectrum", 0) == 1                    # |X|^2 columns
    # (linear columns too, since round 6: the same premise prices every chain -- a normaliser over a handful of nearly equal
    # values, |X|^2 columns -- not only the logarithm)
    scale = {"log": (lambda c: np.log(c)), "db": (lambda c: 20.0 * np.log10(c))}.get(cfg.spectrogramScaling, lambda c: c)
    rng = np.random.default_rng(1)
    moves = np.zeros(len(evals))
    for n, e in enumerate(evals):
        win = cols64[e:e + T]
        amp = np.sqrt(win) if power else win
        with np.errstate(divide="ignore", invalid="ignore"):
            base = o.net_apply(scale(win).reshape(-1), po.F64)
            for _ in range(trials):
                a2 = np.maximum(amp + rng.choice([-1.0, 1.0], size=amp.shape) * delta[e:e + T, None], 1e-300)
                # ... and the scaled column is an fp32 number in the reference itself (vDSP_vdbcon / vvlogf write floats,
                # SyllableDetector.swift:184-212): half an ulp of ITS size is in every value whatever computed it -- 4e-6 on a column
                # of -70 dB, which a network without a normaliser hands on undiminished (round 6, the 6000-draw sweep's draw 3104)
                sc = scale(a2 * a2 if power else a2)
                sc = sc * (1.0 + rng.choice([-1.0, 1.0], size=sc.shape) * 2.0 ** -24)
                out = o.net_apply(sc.reshape(-1), po.F64)
                d = np.abs(out - base) / np.maximum(1.0, np.abs(base))
                moves[n] = max(moves[n], float(np.nan_to_num(d, nan=np.inf).max()))
    return moves


FUSED_KERNELS = ["fused_s_kernel", "fused_r_kernel", "fused_kernel"]


def select_fused(monkeypatch, kernel):
    """The environment under which a handle created next runs `kernel` where several fused kernels take the shape (the switches
    are read once, at syldet_create): the symmetric-fold kernel by default, the register-resident-basis kernel under
    SYLDET_FUSED_NOFOLD=1, the 8-wave kernel under SYLDET_FUSED_CLASSIC=1."""
    monkeypatch.delenv("SYLDET_FUSED_CLASSIC", raising=False)
    monkeypatch.delenv("SYLDET_FUSED_NOFOLD", raising=False)
    if kernel == "fused_r_kernel":
        monkeypatch.setenv("SYLDET_FUSED_NOFOLD", "1")
    elif kernel == "fused_kernel":
        monkeypatch.setenv("SYLDET_FUSED_CLASSIC", "1")
    else:
        assert kernel == "fused_s_kernel"


# ---- sweep bookkeeping: what every random draw needed, written as JSON when the session ends (tests/conftest.py) ---------
SWEEP_LOG = []


def sweep_record(sweep, seed, kernel, err, own, flat_bar, bar_used, reason, widened=None):
    """One (draw, channel) of a random sweep: the worst output error relative to max(1, |anchor|), the fp32 port's own
    distance from the anchor, the flat bar (1e-5 or 4x own; 1e-4 or 30x own for log / dB), the worst ratio of error to the
    bar that was applied, and why a wider bar was applied, if one was ("" | "kappa" | "column level" | "log condition").
    `widened`: what widened_evaluations() found for the evaluations beyond the flat bar."""
    SWEEP_LOG.append({"sweep": sweep, "seed": int(seed), "kernel": kernel, "err": float(err), "own": float(own),
                      "flat_bar": float(flat_bar), "err_over_bar": float(bar_used), "wider_bar": reason, "widened": widened})


def widened_evaluations(errv, own_e, flat, tol, floor_e):
    """The evidence a wider bar needs.  errv: this path's error per evaluation; own_e: the fp32 port's own distance from the
    anchor per evaluation (the reference's operation order in fp32); flat: the flat bar; tol: the bar applied; floor_e: what
    the conditioning argument says NO fp32 evaluation can hold there (kappa 2^-23 for a band that holds 1 / kappa of its
    frames' norm; the anchor's own movement under 2^-23 bin errors for log / dB; None where only the port speaks).
    A bar wider than the flat one is legitimate only where fp32 itself cannot hold the flat bar: the port is already half way
    there (own_e > flat / 2) or the conditioning floor is (floor_e > flat / 2: since round 6 the same half -- the floor is the
    largest movement over a dozen random sign patterns, a sample of what fp32 leaves there, not its worst case; the 6000-draw
    sweep met two evaluations with errors of 1.09e-5 and 1.10e-5 over sampled floors of 8.6e-6 and 6.3e-6).  Anywhere else an
    error above the flat bar is a miss of THIS path's arithmetic, whatever the wider bar says: returned as `unexplained`
    (callers fail on it)."""
    errv = np.asarray(errv, np.float64)
    over = np.nonzero(errv > flat)[0]
    if over.size == 0:
        return None
    own_e = np.broadcast_to(np.asarray(own_e, np.float64), errv.shape)
    tol = np.broadcast_to(np.asarray(tol, np.float64), errv.shape)
    floor = np.broadcast_to(np.asarray(floor_e if floor_e is not None else 0.0, np.float64), errv.shape)
    explained = (own_e[over] > 0.5 * flat) | (floor[over] > 0.5 * flat)
    w = over[np.argmax(errv[over] / flat)]
    rec = {"evaluations_over_flat_bar": int(over.size), "unexplained": int((~explained).sum()),
           "worst": {"evaluation": int(w), "err": float(errv[w]), "own": float(own_e[w]), "fp32_floor": float(floor[w]),
                     "flat_bar": float(flat), "bar_used": float(tol[w])}}
    if (~explained).any():
        u = over[~explained][np.argmax(errv[over][~explained])]
        rec["worst_unexplained"] = {"evaluation": int(u), "err": float(errv[u]), "own": float(own_e[u]), "fp32_floor": float(floor[u]),
                                    "flat_bar": float(flat), "bar_used": float(tol[u])}
    return rec


def check_with_evidence(o: "po.Oracle", cfg, x: np.ndarray, out, fl, w64=None, w32=None, check_flags: bool = True):
    """One channel of one run against the anchor under the ONE rule of the suite (round 6: log / dB columns included): the flat bar
    is 1e-5 (or 4x the fp32 port's own distance from the anchor); an evaluation beyond 1e-5 needs evidence that fp32 itself cannot
    hold it there -- the port beyond half the bar on the evaluations that share a frame with it, or a conditioning floor beyond the
    bar (band_condition behind l2normalize on linear |X| columns, else log_condition's perturbation of the anchor) -- and its bar
    is then 4x / 2x that floor.  Asserts `unexplained == 0`, the values, and the flags outside the guard band; returns
    (worst error, widened record or None)."""
    if w64 is None:
        w64 = o.run(x, po.F64, cfg.rule)[2]
    if w32 is None:
        w32 = o.run(x, po.F32, cfg.rule)[0]
    out = np.asarray(out, np.float64).reshape(w64.shape)
    ok = np.isfinite(w64).all(axis=1)
    assert (np.isfinite(out).all(axis=1) == ok).all(), "NaN/inf evaluations must coincide"
    if not ok.any():
        return 0.0, None
    rel = lambda a: (np.abs(a[ok] - w64[ok]) / np.maximum(1.0, np.abs(w64[ok]))).max(axis=1)
    errv, own_v = rel(out), np.zeros(w64.shape[0])
    own_v[ok] = rel(np.asarray(w32, np.float64))
    T = cfg.timeRange
    own_nb = np.array([own_v[max(0, e - T + 1): e + T].max() for e in range(len(own_v))])[ok]
    flat = max(TOL, 4.0 * float(own_v.max()))
    tol, floor_e = np.full(errv.shape, flat), np.zeros(errv.shape)
    names = [f.function for f in cfg.net.inputProcessing]
    over = np.nonzero(errv > TOL)[0]
    if cfg.spectrogramScaling == "linear" and getattr(cfg, "spectrum", 0) == 0 and names[:1] == ["l2normalize"]:
        floor_e = 2.0 ** -23 * band_condition(o, cfg, x)[ok]
        tol = np.maximum(tol, 4.0 * floor_e)
    elif len(over):
        moves = log_condition(o, cfg, x, o.spectrogram(x, po.F64), np.nonzero(ok)[0][over])
        floor_e[over] = moves
        tol[over] = np.maximum(tol[over], 2.0 * moves)
    wide = widened_evaluations(errv, own_nb, TOL, tol, floor_e)
    assert not wide or wide["unexplained"] == 0, "beyond the flat bar where fp32 holds it: %s" % wide
    assert_outputs_close(out[ok], w64[ok], tol)
    if check_flags and fl is not None:
        assert_flags_exact(np.asarray(fl)[ok], w64[ok], cfg.thresholds, cfg.rule, tol)
        assert not np.asarray(fl)[~ok].any()
    return float(errv.max()), wide


def sweep_summary():
    out = {}
    for r in SWEEP_LOG:
        d = out.setdefault(r["sweep"], {"records": 0, "needed_wider_bar": {}, "worst_err": 0.0, "worst_err_over_own": 0.0,
                                         "worst_err_over_bar": 0.0, "worst_err_over_flat_bar": 0.0, "kernels": {},
                                         "widened_records": []})
        d["records"] += 1
        d["kernels"][r["kernel"]] = d["kernels"].get(r["kernel"], 0) + 1
        d["worst_err"] = max(d["worst_err"], r["err"])
        if r["own"] > 0:
            d["worst_err_over_own"] = max(d["worst_err_over_own"], r["err"] / r["own"])
        d["worst_err_over_bar"] = max(d["worst_err_over_bar"], r["err_over_bar"])
        d["worst_err_over_flat_bar"] = max(d["worst_err_over_flat_bar"], r["err"] / r["flat_bar"])
        if r["err"] > r["flat_bar"]:
            k = r["wider_bar"] or "none (failed)"
            d["needed_wider_bar"][k] = d["needed_wider_bar"].get(k, 0) + 1
        if r["wider_bar"] != "" or r.get("widened"):
            # every record that was held to a bar wider than the flat one, in full: seed, kernel, the errors, the bars, and
            # whether fp32 itself fails the flat bar there (widened_evaluations)
            d["widened_records"].append({k: r[k] for k in ("seed", "kernel", "err", "own", "flat_bar", "err_over_bar", "wider_bar", "widened")})
    return out


def launched(det):
    """Names of the kernels of the detector's last batch call, in launch order -- without the exact recomputation behind a fused
    kernel ("fixup_kernel"), which syldet_timings lists exactly for the calls that gave it work (level steps inside a pass-scaled
    kernel's pass do; tests/test_extremes_gpu.py holds that rule itself)."""
    return [name for name, _ in det.lastTimings() if name != "fixup_kernel"]
