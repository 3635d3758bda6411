#!/usr/bin/env python3
"""Debug: WHERE a wrong block of the wide GEMM went wrong.  Needs a library built with -DSYLDET_WIDE_X_TRACE (tools/variant_libs.sh;
SYLDET_LIB points at it): the GEMM then stores evaluation tile 0's running output sum of every lane after every chunk
([workgroup][wave][chunk][lane]).  The trace of the shipped order is taken first (SYLDET_WIDE_NOSTAGGER=1; twice: it must repeat),
then the form under test runs N times; for the first wrong (workgroup, wave) pairs the first deviating chunk is printed with the
per-lane-group changes, against the shipped order's own steps of the chunks around it and against the host model's split of the
step into unit tiles (tools/debug/wide_model.py).

    SYLDET_LIB=.../lib_x_trace/libsyldet.so python tools/debug/wide_trace.py [--runs 4] [--show 8] [--C 16]"""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import syllable_detector_swift_amd as sd
from syllable_detector_swift_amd import synth, _abi
import wide_model as wm

np.set_printoptions(precision=5, linewidth=220, suppress=True)
ap = argparse.ArgumentParser()
ap.add_argument("--runs", type=int, default=4)
ap.add_argument("--show", type=int, default=8)
ap.add_argument("--C", type=int, default=16)
ap.add_argument("--S", type=int, default=1 << 23)
a = ap.parse_args()
hip = ctypes.CDLL("libamdhip64.so")
x = synth.channels_on_device(a.C, a.S, torch.device("cuda", 0), fs=wm.cfg.samplingRate)
NCH = wm.NCH


def fetch_trace(det, n_wg):
    p, n = ctypes.c_void_p(), ctypes.c_size_t()
    _abi.lib.syldet_debug_wide_trace(ctypes.byref(p), ctypes.byref(n))
    want = n_wg * 8 * NCH * 5 * 64 * 4
    assert p.value and n.value >= want * 4, "no trace: is SYLDET_LIB a -DSYLDET_WIDE_X_TRACE build?"
    out = np.empty(want, np.uint32)
    assert hip.hipMemcpy(out.ctypes.data_as(ctypes.c_void_p), p, ctypes.c_size_t(want * 4), 2) == 0
    return out.reshape(n_wg, 8, NCH, 5, 64, 4)          # [workgroup][wave][chunk][record][lane][4] as bits; records: tile 0's hidden values of unit tiles 0, 1; the second-layer weights read for them; the running sums (tile 0, tile 1, -, -)


os.environ["SYLDET_WIDE_NOSTAGGER"] = "1"            # the reference: round 4's order, every wave of a workgroup in the same chunk
with sd.SyllableDetector(wm.cfg, channels=a.C, engine=_abi.ENGINE_WIDE_BF16) as det:
    E = det.countEvaluations(a.S)
    wg_per_channel = (E + 255) // 256
    n_wg = wg_per_channel * a.C
    out_ref = det.run(x)[0][..., 0].cpu().numpy().copy(); torch.cuda.synchronize()
    ref = fetch_trace(det, n_wg).copy()
    det.run(x); torch.cuda.synchronize()
    again = fetch_trace(det, n_wg)
    if not (again == ref).all():
        ne0 = again != ref
        print("the shipped order's trace differs run to run: records %s; first %s" % ([int(ne0[:, :, :, r].sum()) for r in range(5)], np.argwhere(ne0)[:5].tolist()))
        w = np.argwhere(ne0)[0]
        print("   values", again[tuple(w[:5])].view(np.float32), ref[tuple(w[:5])].view(np.float32))
    cols = det.spectrogram(x)
    torch.cuda.synchronize()
print("shipped order: %d workgroups, trace repeats" % n_wg, flush=True)
os.environ.pop("SYLDET_WIDE_NOSTAGGER")

shown = 0
with sd.SyllableDetector(wm.cfg, channels=a.C, engine=_abi.ENGINE_WIDE_BF16) as det:
    for k in range(a.runs):
        out = det.run(x)[0][..., 0].cpu().numpy(); torch.cuda.synchronize()
        wrong_out = np.argwhere(out != out_ref)
        wrong_blocks = sorted({(int(c), int(e) // 16) for c, e in wrong_out})
        print("run %d: %d blocks of 16 outputs differ from the shipped order's: %s" % (k, len(wrong_blocks), [(c, b // 16, (b % 16) // 2, b % 2) for c, b in wrong_blocks[:6]]), "(channel, workgroup, wave, tile)", flush=True)
        bad = fetch_trace(det, n_wg)
        ne = bad != ref                                                                  # [wg][wave][chunk][record][lane][4]
        names = ["hidden ut0", "hidden ut1", "w1 ut0", "w1 ut1", "sums"]
        print("   records that differ anywhere: " + ", ".join("%s %d" % (names[r], int(ne[:, :, :, r].any(-1).any(-1).sum())) for r in range(5)), flush=True)
        ysum_ne = ne[:, :, :, 4, :, 0].any(-1)                                           # tile 0's running sum, [wg][wave][chunk]
        for wg, wave in np.argwhere(ysum_ne.any(-1)):
            if shown >= a.show:
                break
            shown += 1
            c = int(np.argmax(ysum_ne[wg, wave]))
            ch, blk = divmod(int(wg), wg_per_channel)
            f = lambda r_: bad[wg, wave, c, r_].view(np.float32)
            prev = bad[wg, wave, c - 1, 4].view(np.float32)[:, 0] if c else np.zeros(64, np.float32)
            got, want = f(4)[:, 0], ref[wg, wave, c, 4].view(np.float32)[:, 0]
            lanes = np.nonzero(got != want)[0]
            print("  channel %d workgroup %d wave %d: tile 0's sum first deviates after chunk %d (buffer %d) in %d lanes (by lane group %s); tile 1's sum %s; hidden values and weights of that chunk %s" % (
                ch, blk, wave, c, c % 3, len(lanes), np.bincount(lanes >> 4, minlength=4).tolist(),
                "deviates too" if ne[wg, wave, c, 4, :, 1].any() else "is right", "differ" if ne[wg, wave, c, :4].any() else "are the shipped order's"))
            # replay the eight multiply-adds of the chunk in fp32 from the traced values, with every single term left out / every prefix only
            h = np.concatenate([f(0), f(1)], 1).astype(np.float32)                        # [lane][8]: ut0 j0..3, ut1 j0..3
            w = np.concatenate([f(2), f(3)], 1).astype(np.float32)
            def chain(terms, start):
                y = start.astype(np.float32).copy()
                for k_ in terms:
                    y = (h[:, k_].astype(np.float64) * w[:, k_].astype(np.float64) + y.astype(np.float64)).astype(np.float32)
                return y
            full = chain(range(8), prev)
            print("     replay of all eight terms from the traced values == shipped sum: %s" % bool((full[lanes] == want[lanes]).all()))
            found = False
            for k_ in range(8):
                if (chain([q for q in range(8) if q != k_], prev)[lanes] == got[lanes]).all():
                    print("     == the sum WITHOUT term %d (unit tile %d, j %d)" % (k_, k_ // 4, k_ % 4)); found = True
                if (chain(range(k_), prev)[lanes] == got[lanes]).all():
                    print("     == the sum of the first %d terms only" % k_); found = True
                if (chain(range(k_, 8), prev)[lanes] == got[lanes]).all() and k_:
                    print("     == the sum of terms %d.. only (on the previous sum)" % k_); found = True
                if c and (chain(range(k_, 8), bad[wg, wave, c - 2, 4].view(np.float32)[:, 0] if c > 1 else np.zeros(64, np.float32))[lanes] == got[lanes]).all():
                    print("     == terms %d.. added to the sum after chunk %d (the chunk before's terms lost)" % (k_, c - 2)); found = True
            if not found:
                l = lanes[0]
                print("     no single-term explanation; lane %d: previous %.9g got %.9g want %.9g  h %s w %s" % (l, prev[l], got[l], want[l], h[l], w[l]))
print("done")
